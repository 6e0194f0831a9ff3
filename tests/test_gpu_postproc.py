"""GPU parity of nms / fast-NMS / greedy NMS / mask assembly / after_nms against the CPU oracle and the
reference's golden vectors.  Bar: ids, scores, coefs, BOXES and pixel boxes bit-exact against the oracle; binary masks may
differ only where the interpolated value is within 1e-4 of the 0.5 threshold.

The one transcendental of the path, the decode's exp (utils/output_utils.py:150), is MKL VML inside the reference's torch
build: closed source, host-ISA dependent, 1 ulp off the correctly rounded value in 1.1 % of inputs
(oracle/make_golden_exp.py, tests/test_oracle_expf.py).  The oracle is therefore run with exp='cr' (exp rounded to nearest,
oracle/expf_cr.c), which the kernel matches on every bit; against the reference's frozen outputs the boxes may differ by
that one ulp of exp in a few coordinates (asserted: >= 99 % of coordinates equal, the rest within 1 ulp of the box size), while ids,
scores, coefs, pixel boxes and masks of the goldens are reproduced exactly, end to end, from the kernel's own boxes."""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import yolact_ref as R
from yolact_minimal_amd.config import build_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _cfg(**kw):
    cfg = build_cfg('res101_coco', 'val', 544)
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _gpu_nms(cls, box, coef, proto, anchors, cfg):
    from yolact_minimal_amd.utils.output_utils import nms
    return nms(cls.to(DEV), box.to(DEV), coef.to(DEV), proto.to(DEV), anchors.to(DEV), cfg)


def _check_nms(g, r):
    """Everything nms returns, bit for bit, against the oracle run with the correctly rounded exp."""
    assert (g[0] is None) == (r[0] is None)
    if r[0] is None:
        return
    assert g[0].dtype == torch.int64
    np.testing.assert_array_equal(g[0].cpu().numpy(), r[0].numpy())
    np.testing.assert_array_equal(g[1].cpu().numpy(), r[1].numpy())
    np.testing.assert_array_equal(g[2].cpu().numpy(), r[2].numpy())
    np.testing.assert_array_equal(g[3].cpu().numpy(), r[3].numpy())


def _check_boxes_vs_reference_exp(boxes, golden_boxes):
    """Against the reference's frozen boxes (decode through MKL's exp): equal except for its 1-ulp deviations."""
    b = boxes.cpu().numpy()
    assert float((b == golden_boxes).mean()) >= 0.99
    np.testing.assert_allclose(b, golden_boxes, rtol=0, atol=1.2e-7)     # 1 ulp of w or h (< 1), halved, re-added


def test_expf_cr_kernel_is_bit_identical_to_the_oracle():
    from yolact_minimal_amd import hip
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(2_000_000, generator=g) * 0.5, (torch.rand(2_000_000, generator=g) * 2 - 1) * 110.0,
                   torch.tensor([float('nan'), float('inf'), -float('inf'), 0.0, -0.0, 88.72, 88.73, 89.0, 89.5, -87.4, -103.0,
                                 -103.98, -104.0, -104.5, -1e30, 1e30, 1e-40, -1e-40])])
    got = hip.expf_cr(x.to(DEV)).cpu()
    want = R.expf_cr(x)
    np.testing.assert_array_equal(got.numpy().view(np.int32), want.numpy().view(np.int32))


def _check_after(ga, r, h, w):
    from_oracle = R.after_nms(r[0], r[1], r[2], r[3], r[4], h, w, return_soft=True)
    ids, scores, boxes_px, masks, soft, up = from_oracle
    assert ga[2].dtype == torch.int32 and tuple(ga[3].shape) == (ids.numel(), h, w)
    np.testing.assert_array_equal(ga[2].cpu().numpy(), boxes_px.numpy())
    gm = ga[3].cpu()
    assert set(torch.unique(gm).tolist()) <= {0.0, 1.0}
    diff = gm != masks
    if bool(diff.any()):
        # every disagreeing pixel must sit on the threshold
        assert float((up[diff] - 0.5).abs().max()) < 1e-4, f'{int(diff.sum())} mask pixels differ away from 0.5'
    assert float(diff.float().mean()) < 1e-4


@pytest.mark.parametrize('tag', ['small128', 'empty128', 'degenerate128'])
def test_nms_small_goldens(golden_dir, tag):
    from yolact_minimal_amd.utils.output_utils import after_nms
    g = np.load(os.path.join(golden_dir, f'post_{tag}.npz'))
    cls, box, coef, proto = (torch.from_numpy(g[k]) for k in ('in_class', 'in_box', 'in_coef', 'in_proto'))
    anchors = torch.from_numpy(g['in_anchors'])
    cfg = _cfg(img_size=128)
    out = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    if int(g['n']) == 0:
        assert out == (None, None, None, None, None)
        assert after_nms(*out, 64, 64) == (None, None, None, None)
        return
    # against the reference's own outputs
    np.testing.assert_array_equal(out[0].cpu().numpy(), g['ids'])
    np.testing.assert_array_equal(out[1].cpu().numpy(), g['scores'])
    _check_boxes_vs_reference_exp(out[2], g['boxes'])
    np.testing.assert_array_equal(out[3].cpu().numpy(), g['coefs'])
    r = R.nms(cls, box, coef, proto, anchors, exp='cr')
    _check_nms(out, r)
    for key in g.files:
        if key.startswith('px_boxes_'):
            h, w = (int(v) for v in key[len('px_boxes_'):].split('x'))
            boxes_in = out[2].clone()                            # the kernel's OWN boxes, end to end
            ga = after_nms(out[0], out[1], boxes_in, out[3], out[4], h, w, cfg)
            np.testing.assert_array_equal(ga[2].cpu().numpy(), g[key])                           # the reference's pixel boxes
            np.testing.assert_array_equal(boxes_in.cpu().numpy(), r[2].numpy() * max(h, w))      # in-place scaling
            packed = np.packbits(ga[3].cpu().numpy().astype(np.uint8).reshape(-1))
            mism = int(np.unpackbits(packed ^ g[f'masks_{h}x{w}_packed']).sum())
            assert mism <= max(2, int(1e-5 * ga[3].numel())), f'{mism} mask pixels differ from the reference'
            _check_after(ga, (r[0], r[1], r[2].clone(), r[3], r[4]), h, w)


def test_nms_ties_against_stable_oracle(golden_dir):
    """torch.sort is unstable for n > 16, so the reference's order among equal scores is implementation
    defined; the kernel resolves ties as a stable sort would.  Scores must still match the reference."""
    g = np.load(os.path.join(golden_dir, 'post_ties128.npz'))
    cls, box, coef, proto = (torch.from_numpy(g[k]) for k in ('in_class', 'in_box', 'in_coef', 'in_proto'))
    anchors = torch.from_numpy(g['in_anchors'])
    out = _gpu_nms(cls, box, coef, proto, anchors, _cfg(img_size=128))
    np.testing.assert_array_equal(out[1].cpu().numpy(), g['scores'])
    r = R.nms(cls, box, coef, proto, anchors, stable=True, exp='cr')
    _check_nms(out, r)


@pytest.mark.parametrize('ncls,top_k,max_det', [(3, 200, 100), (64, 200, 100), (65, 50, 7), (130, 200, 128), (200, 200, 100), (255, 17, 100)])
def test_nms_class_counts_and_limits(ncls, top_k, max_det):
    """The final select is instantiated per 64 classes (1..4 lists per lane group) and the reference's limits are config values
    (config.py: top_k, max_detections): every instantiation, cut-offs below / at the caps, scores quantised so that ties cross
    class boundaries and sit on the max_det cut -- against the oracle with a stable sort, bit for bit."""
    gen = torch.Generator().manual_seed(1000 + ncls)
    anchors = R.anchors_for(128, [24, 48, 96, 192, 384])
    n = anchors.shape[0]
    logits = torch.randn(1, n, ncls + 1, generator=gen) * 3.0
    logits[..., 0] += 1.0
    cls = torch.round(torch.softmax(logits, -1) * 32.0) / 32.0            # 33 score levels: ties everywhere
    box = torch.randn(1, n, 4, generator=gen) * 0.5
    coef = torch.tanh(torch.randn(1, n, 32, generator=gen))
    proto = torch.randn(1, 32, 32, 32, generator=gen)
    cfg = _cfg(img_size=128, top_k=top_k, max_detections=max_det)
    out = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    r = R.nms(cls, box, coef, proto, anchors, top_k=top_k, max_det=max_det, img_size=128, stable=True, exp='cr')
    assert r[0] is not None and r[0].numel() == min(max_det, r[0].numel())
    _check_nms(out, r)


@pytest.mark.parametrize('seed,bg,tag', [(1, 4.0, 'dense544'), (2, 9.0, 'sparse544')])
def test_nms_full_size(golden_dir, seed, bg, tag):
    """BASELINE full geometry: 18 525 anchors, 136x136x32 prototypes, ~17.8k / ~300 candidates."""
    from yolact_minimal_amd.utils.output_utils import after_nms
    g = np.load(os.path.join(golden_dir, f'post_{tag}.npz'))
    cls, box, coef, proto = R.synth_head_outputs(18525, seed=seed, bg_bias=bg)
    anchors = R.anchors_for(544, [24, 48, 96, 192, 384])
    cfg = _cfg()
    out = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    # the inputs are re-generated on this host (softmax/tanh of the CPU libm may differ in the last ulp from
    # the build container's), so the golden comparison of VALUES carries 1-ulp slack; the comparison with the
    # oracle run on the very same inputs (below) is exact.
    np.testing.assert_array_equal(out[0].cpu().numpy(), g['ids'])
    np.testing.assert_allclose(out[1].cpu().numpy(), g['scores'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(out[2].cpu().numpy(), g['boxes'], rtol=0, atol=2e-7)
    np.testing.assert_allclose(out[3].cpu().numpy(), g['coefs'], rtol=0, atol=1.2e-7)
    r = R.nms(cls, box, coef, proto, anchors, exp='cr')
    _check_nms(out, r)
    for key in g.files:
        if key.startswith('px_boxes_'):
            h, w = (int(v) for v in key[len('px_boxes_'):].split('x'))
            ga = after_nms(out[0], out[1], out[2].clone(), out[3], out[4], h, w, cfg)     # the kernel's own boxes
            np.testing.assert_array_equal(ga[2].cpu().numpy(), g[key])
            _check_after(ga, (r[0], r[1], r[2].clone(), r[3], r[4]), h, w)
            np.testing.assert_allclose(ga[3].sum(dim=(1, 2)).cpu().numpy(), g[f'masks_{h}x{w}_area'], rtol=0, atol=3)
            packed = np.packbits(ga[3].cpu().numpy().astype(np.uint8).reshape(-1))
            mism = int(np.unpackbits(packed ^ g[f'masks_{h}x{w}_packed']).sum())
            assert mism <= int(1e-5 * ga[3].numel()), f'{mism} mask pixels differ from the reference'


def test_fast_nms_properties_full_size():
    """Size-independent properties at BASELINE size: scores sorted, ids in range, <= max_det, every kept box
    has IoU <= 0.5 with every higher-scored kept box of its class, idempotent under a second run."""
    cls, box, coef, proto = R.synth_head_outputs(18525, seed=9)
    anchors = R.anchors_for(544, [24, 48, 96, 192, 384])
    cfg = _cfg()
    a = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    b = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)
    ids, sc, bx = a[0].cpu(), a[1].cpu(), a[2].cpu()
    assert ids.numel() <= cfg.max_detections and int(ids.min()) >= 0 and int(ids.max()) < 80
    assert bool((sc[:-1] >= sc[1:]).all())
    iou = R.pairwise_iou(bx[None], bx[None])[0]
    same = ids[:, None] == ids[None, :]
    upper = torch.triu(torch.ones_like(iou, dtype=torch.bool), 1)
    assert float(iou[same & upper].max()) <= 0.5


def test_traditional_nms_matches_oracle():
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=3, bg_bias=5.0)
    a128 = R.anchors_for(128, [int(128 / 544 * s) for s in (24, 48, 96, 192, 384)])
    cfg = _cfg(img_size=128, traditional_nms=True)
    out = _gpu_nms(cls, box, coef, proto, a128, cfg)
    r = R.nms(cls, box, coef, proto, a128, traditional=True, img_size=128, stable=True, exp='cr')
    _check_nms(out, r)
    # full size, sparse (greedy is O(n^2) per class)
    cls, box, coef, proto = R.synth_head_outputs(18525, seed=2, bg_bias=9.0)
    anchors = R.anchors_for(544, [24, 48, 96, 192, 384])
    cfg = _cfg(traditional_nms=True)
    out = _gpu_nms(cls, box, coef, proto, anchors, cfg)
    r = R.nms(cls, box, coef, proto, anchors, traditional=True, img_size=544, stable=True, exp='cr')
    _check_nms(out, r)


def test_greedy_nms_drop_in_known_answers():
    """ym_greedy_nms == cython_nms.nms on the hand-derived cases + random data vs the C oracle."""
    from yolact_minimal_amd import hip

    def run(dets, thr):
        d = torch.from_numpy(np.ascontiguousarray(dets, np.float32)).to(DEV)
        n = d.shape[0]
        keep = torch.zeros(max(n, 1), dtype=torch.uint8, device=DEV)
        cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
        ws = torch.empty(hip.lib().ym_greedy_nms_workspace_bytes(n), dtype=torch.uint8, device=DEV)
        rc = hip.lib().ym_greedy_nms(ctypes.c_void_p(d.data_ptr()) if n else None, n, thr,
                                     ctypes.c_void_p(keep.data_ptr()), ctypes.c_void_p(cnt.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), ws.numel(), hip.stream_ptr())
        hip.check(rc, 'ym_greedy_nms')
        idx = torch.nonzero(keep[:n]).flatten().cpu().tolist()
        assert int(cnt.item()) == len(idx)
        return idx

    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [20, 20, 29, 29, 0.7]], np.float32)
    assert run(dets, 0.5) == [0, 2]
    assert run(dets, 0.51) == [0, 1, 2]
    assert run(np.array([[0, 0, 9, 9, 0.9], [4, 0, 13, 9, 0.8], [8, 0, 17, 9, 0.7]], np.float32), 0.4) == [0, 2]
    assert run(np.zeros((0, 5), np.float32), 0.5) == []
    rng = np.random.default_rng(0)
    for n in (1, 7, 64, 1500):
        xy = rng.uniform(0, 500, (n, 2)).astype(np.float32)
        wh = rng.uniform(5, 120, (n, 2)).astype(np.float32)
        dets = np.concatenate([xy, xy + wh, rng.uniform(0.05, 1, (n, 1)).astype(np.float32)], 1)
        assert run(dets, 0.5) == R.greedy_nms(dets, 0.5).tolist()


def test_mask_assemble_matches_oracle():
    from yolact_minimal_amd import hip
    g = torch.Generator().manual_seed(4)
    for (hp, wp, n) in ((136, 136, 100), (32, 32, 7), (17, 23, 33)):
        proto = torch.relu(torch.randn(hp, wp, 32, generator=g))
        coef = torch.tanh(torch.randn(n, 32, generator=g))
        xy = torch.rand(n, 2, generator=g) * 0.7
        boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 0.3], 1)
        boxes[0] = torch.tensor([0.9, 0.2, 0.1, 0.8])      # x1 > x2: sanitize swaps
        out = torch.empty(n, hp, wp, device=DEV)
        hip.mask_assemble(proto.to(DEV), coef.to(DEV), boxes.to(DEV), out, True)
        want = R.assemble_masks(proto, coef, boxes, True)
        torch.testing.assert_close(out.cpu(), want, rtol=0, atol=1e-4)     # north_star: mask within 1e-4
        assert torch.equal(out.cpu() == 0, want == 0)                      # crop window is bit-exact
        hip.mask_assemble(proto.to(DEV), coef.to(DEV), boxes.to(DEV), out, False)
        torch.testing.assert_close(out.cpu(), R.assemble_masks(proto, coef, boxes, False), rtol=0, atol=1e-4)


def test_after_nms_visual_thre_and_none():
    from yolact_minimal_amd.utils.output_utils import after_nms
    assert after_nms(None, None, None, None, None, 10, 10) == (None, None, None, None)
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=3, bg_bias=5.0)
    a128 = R.anchors_for(128, [int(128 / 544 * s) for s in (24, 48, 96, 192, 384)])
    cfg = _cfg(img_size=128)
    out = _gpu_nms(cls, box, coef, proto, a128, cfg)
    cfg.visual_thre, cfg.save_lincomb, cfg.no_crop = float(out[1][10]), False, False
    ga = after_nms(out[0], out[1], out[2].clone(), out[3], out[4], 50, 70, cfg)
    assert ga[0].numel() == int((out[1] >= cfg.visual_thre).sum())
    cfg.visual_thre = 2.0
    assert after_nms(out[0], out[1], out[2].clone(), out[3], out[4], 50, 70, cfg) == (None, None, None, None)


@pytest.mark.parametrize('hw', [(480, 640), (544, 544), (300, 200)])
def test_batched_postprocessing_equals_per_image(hw):
    """`nms_batch` + `after_nms_batch` (one launch set for the batch, ONE host read) against the batch-1 calls image by image:
    everything bit-identical — including an image without detections and one with fewer candidates than top_k.
    (300, 200) takes the two-kernel fallback of ym_after_nms_batch (image much smaller than 2.7x the prototype map)."""
    from yolact_minimal_amd.utils.output_utils import nms, after_nms, nms_batch, after_nms_batch
    h, w = hw
    cfg = _cfg()
    anchors = R.anchors_for(544, [24, 48, 96, 192, 384])
    parts = [R.synth_head_outputs(18525, seed=1), R.synth_head_outputs(18525, seed=2, bg_bias=9.0),
             R.synth_head_outputs(18525, seed=4, bg_bias=30.0), R.synth_head_outputs(18525, seed=5, bg_bias=7.5),
             R.synth_head_outputs(18525, seed=7, bg_bias=5.0)]
    cls, box, coef, proto = (torch.cat([p[i] for p in parts], 0).to(DEV) for i in range(4))
    per_image = []
    for b in range(len(parts)):
        r = nms(cls[b:b + 1], box[b:b + 1], coef[b:b + 1], proto[b:b + 1], anchors.to(DEV), cfg)
        a = after_nms(r[0], r[1], r[2].clone() if r[2] is not None else None, r[3], r[4], h, w, cfg)
        per_image.append((r, a))
    dets = nms_batch(cls, box, coef, proto, anchors.to(DEV), cfg)
    split = dets.split()
    assert [s[0] is None for s in split] == [False, False, True, False, False]
    for (r, _), s in zip(per_image, split):
        assert (r[0] is None) == (s[0] is None)
        if r[0] is not None:
            for x, y in zip(r[:4], s[:4]):
                assert torch.equal(x, y)
    got = after_nms_batch(dets, h, w, cfg)
    for (_, a), g_ in zip(per_image, got):
        assert (a[0] is None) == (g_[0] is None)
        if a[0] is not None:
            assert g_[2].dtype == torch.int32 and g_[3].shape == a[3].shape
            for x, y in zip(a, g_):
                assert torch.equal(x, y)


def test_fused_after_nms_matches_the_two_kernel_path():
    """ym_after_nms_batch's fused kernel (soft masks only in LDS, zero-filled tiles outside the crop window) against
    ym_mask_assemble + ym_mask_resize_binarize: identical except where the interpolated value sits within 1e-6 of the threshold
    (the fused dot product sums in k order, the MFMA in its own order)."""
    from yolact_minimal_amd import hip
    from yolact_minimal_amd.utils.output_utils import after_nms
    g = torch.Generator().manual_seed(8)
    for (hp, n, h, w, crop) in ((136, 100, 480, 640, True), (136, 37, 544, 544, True), (136, 9, 427, 640, False), (32, 20, 96, 128, True),
                                (34, 5, 333, 517, True)):
        proto = torch.relu(torch.randn(hp, hp, 32, generator=g)).to(DEV)
        coef = torch.tanh(torch.randn(n, 32, generator=g)).to(DEV)
        xy = torch.rand(n, 2, generator=g) * 0.7
        boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 0.3], 1)
        boxes[0] = torch.tensor([0.9, 0.2, 0.1, 0.8])
        boxes[-1] = torch.tensor([0.0, 0.0, 1.0, 1.0])
        boxes = boxes.to(DEV)
        soft = torch.empty(n, hp, hp, device=DEV)
        hip.mask_assemble(proto, coef, boxes, soft, crop)
        want = torch.empty(n, h, w, device=DEV)
        hip.mask_resize_binarize(soft, h, w, want)
        cfg = _cfg(no_crop=not crop)
        ids = torch.zeros(n, dtype=torch.int64, device=DEV)
        sc = torch.ones(n, device=DEV)
        got = after_nms(ids, sc, boxes.clone(), coef, proto, h, w, cfg)[3]
        diff = got != want
        if bool(diff.any()):
            up = torch.nn.functional.interpolate(soft[None], (max(h, w), max(h, w)), mode='bilinear', align_corners=False)[0][:, :h, :w]
            assert float((up[diff] - 0.5).abs().max()) < 1e-5
        assert float(diff.float().mean()) < 1e-5
