"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference) in the
build container, and pins the CPU restatement (oracle/yolact_ref.py) against it.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python oracle/make_golden.py
The reference cannot travel to the GPU box; only the vectors written here do.

Import shim: `utils/output_utils.py` imports cv2 and cython_nms at module top and
neither exists in this image, so empty stand-in *modules* are registered before the import (the
functions exercised here — nms, fast_nms, after_nms — never touch them); `config.py` creates result
directories in the cwd at import time, so the import happens from a temp dir.
"""
import argparse
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from oracle import yolact_ref as R  # noqa: E402


def import_reference():
    os.chdir(tempfile.mkdtemp(prefix='yolact_ref_cwd_'))
    sys.path.insert(0, '/root/reference')
    for name in ('cv2', 'cython_nms'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['cython_nms'].nms = None
    ref_config = importlib.import_module('config')
    ref_yolact = importlib.import_module('modules.yolact')
    ref_out = importlib.import_module('utils.output_utils')
    ref_box = importlib.import_module('utils.box_utils')
    return ref_config, ref_yolact, ref_out, ref_box


def ref_cfg(ref_config, name, img_size, mode='val'):
    a = argparse.Namespace(cfg=name, img_size=img_size, weight=None, traditional_nms=False, val_num=-1,
                           coco_api=False, resume=None, train_bs=8, bs_per_gpu=8, val_interval=4000)
    a.mode, a.cuda, a.gpu_id = mode, False, None
    return getattr(ref_config, name)(a)


def tensor_digest(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def gen_state(ref_config, ref_yolact):
    """Seeded construction: per-tensor digests of the reference's state dict."""
    out = {}
    for name, seed in (('res50_coco', 3), ('res101_coco', 5)):
        cfg = ref_cfg(ref_config, name, 64)
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg)
        sd = net.state_dict()
        keys = list(sd.keys())
        out[f'{name}_keys'] = np.array(keys)
        out[f'{name}_digest'] = np.stack([tensor_digest(sd[k].float()) for k in keys])
        out[f'{name}_seed'] = np.array(seed)
        out[f'{name}_anchors64'] = np.array(net.anchors, dtype=np.float64)
    cfg = ref_cfg(ref_config, 'res101_coco', 544)
    out['anchors544_f32'] = torch.tensor(_anchors_only(ref_yolact, cfg)).reshape(-1, 4).numpy()
    np.savez_compressed(os.path.join(OUT, 'state.npz'), **out)
    print('state.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


def _anchors_only(ref_yolact, cfg):
    import math
    from utils.box_utils import make_anchors
    anchors = []
    for i, size in enumerate([math.ceil(cfg.img_size / s) for s in (8, 16, 32, 64, 128)]):
        anchors += make_anchors(cfg, size, size, cfg.scales[i])
    return anchors


def gen_forward(ref_config, ref_yolact):
    """Eval forward on small images: full output tensors; 544: digests + strided samples."""
    cases = [('res50_coco', 64, 1, 21), ('res50_coco', 96, 2, 22), ('res101_coco', 128, 1, 23)]
    for name, size, batch, seed in cases:
        cfg = ref_cfg(ref_config, name, size)
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).eval()
        sd = net.state_dict()
        R.randomize_bn_(sd, seed + 100)
        R.randomize_bias_(sd, seed + 200)
        net.load_state_dict(sd)
        img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
        with torch.no_grad():
            ref = net(img)
            mine = R.forward_eval(img, sd)
        for a, b in zip(ref, mine):
            assert torch.equal(a, b), f'oracle restatement differs from the reference ({name}@{size})'
        np.savez_compressed(os.path.join(OUT, f'forward_{name}_{size}_b{batch}.npz'),
                            seed=np.array(seed), class_pred=ref[0].numpy(), box_pred=ref[1].numpy(),
                            coef_pred=ref[2].numpy(), proto_out=ref[3].numpy(),
                            img_digest=tensor_digest(img))
        print(f'forward {name}@{size} b{batch}: ok, max|class|={ref[0].max():.4f} proto max={ref[3].max():.3f}')

    # full-size digest (the "550-class" config really is 544: SURVEY.md §0.1)
    for name, seed in (('res50_coco', 31), ('res101_coco', 32)):
        cfg = ref_cfg(ref_config, name, 544)
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).eval()
        sd = net.state_dict()
        R.randomize_bn_(sd, seed + 100)
        R.randomize_bias_(sd, seed + 200)
        net.load_state_dict(sd)
        img = torch.randn(1, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300))
        with torch.no_grad():
            ref = net(img)
            mine = R.forward_eval(img, sd)
        for a, b in zip(ref, mine):
            assert torch.equal(a, b)
        np.savez_compressed(
            os.path.join(OUT, f'forward_{name}_544_digest.npz'), seed=np.array(seed),
            class_digest=tensor_digest(ref[0]), box_digest=tensor_digest(ref[1]),
            coef_digest=tensor_digest(ref[2]), proto_digest=tensor_digest(ref[3]),
            class_sample=ref[0][0, ::37].numpy(), box_sample=ref[1][0, ::37].numpy(),
            coef_sample=ref[2][0, ::37].numpy(), proto_sample=ref[3][0, ::5, ::5].numpy(),
            img_digest=tensor_digest(img))
        print(f'forward {name}@544 digest ok')


class _Cfg:
    def __init__(self, **kw):
        self.traditional_nms = False
        self.nms_score_thre = 0.05
        self.nms_iou_thre = 0.5
        self.top_k = 200
        self.max_detections = 100
        self.img_size = 544
        self.visual_thre = 0.0
        self.save_lincomb = False
        self.no_crop = False
        self.__dict__.update(kw)


def _nms_case(ref_out, tag, cls, box, coef, proto, anchors, img_hw, store_inputs, cfg=None):
    cfg = cfg or _Cfg()
    with torch.no_grad():
        r = ref_out.nms(cls.clone(), box.clone(), coef.clone(), proto.clone(), anchors, cfg)
        m = R.nms(cls, box, coef, proto, anchors, cfg.nms_score_thre, cfg.nms_iou_thre, cfg.top_k,
                  cfg.max_detections)
    out = {}
    if store_inputs:
        out.update(in_class=cls.numpy(), in_box=box.numpy(), in_coef=coef.numpy(), in_proto=proto.numpy(),
                   in_anchors=anchors.numpy())
    else:
        out.update(in_class_digest=tensor_digest(cls), in_box_digest=tensor_digest(box),
                   in_coef_digest=tensor_digest(coef), in_proto_digest=tensor_digest(proto))
    if r[0] is None:
        assert m[0] is None
        out['n'] = np.array(0)
    else:
        for a, b in zip(r[:4], m[:4]):
            # bit-equal; NaNs (torch.equal is False for them) must sit in the same places and everything else must still match
            same = torch.equal(a, b) or (a.is_floating_point() and torch.equal(torch.isnan(a), torch.isnan(b)) and
                                         torch.equal(torch.nan_to_num(a, nan=0.0), torch.nan_to_num(b, nan=0.0)))
            assert same, f'nms restatement differs ({tag})'
        out.update(n=np.array(r[0].numel()), ids=r[0].numpy(), scores=r[1].numpy(), boxes=r[2].numpy(),
                   coefs=r[3].numpy())
        for (h, w) in img_hw:
            with torch.no_grad():
                ra = ref_out.after_nms(r[0], r[1], r[2].clone(), r[3], r[4], h, w, cfg)
                ma = R.after_nms(m[0], m[1], m[2], m[3], m[4], h, w, return_soft=True)
            assert torch.equal(ra[2], ma[2]) and torch.equal(ra[3], ma[3]), f'after_nms restatement differs ({tag})'
            out[f'px_boxes_{h}x{w}'] = ra[2].numpy()
            out[f'masks_{h}x{w}_packed'] = np.packbits(ra[3].numpy().astype(np.uint8).reshape(-1))
            out[f'masks_{h}x{w}_area'] = ra[3].sum(dim=(1, 2)).numpy()
    np.savez_compressed(os.path.join(OUT, f'post_{tag}.npz'), **out)
    print(f'post {tag}: n={int(out["n"])}')


def gen_post(ref_config, ref_yolact, ref_out):
    anchors544 = R.anchors_for(544, [24, 48, 96, 192, 384])
    # (1) dense worst case at the full 544 geometry: ~17.8k candidates -> 100 detections
    cls, box, coef, proto = R.synth_head_outputs(18525, seed=1)
    _nms_case(ref_out, 'dense544', cls, box, coef, proto, anchors544, [(480, 640), (544, 544)], False)
    # (2) sparse: strong background -> a few hundred candidates
    cls, box, coef, proto = R.synth_head_outputs(18525, seed=2, bg_bias=9.0)
    _nms_case(ref_out, 'sparse544', cls, box, coef, proto, anchors544, [(300, 200)], False)
    # (3) small geometry with stored inputs (img 128: 1023 anchors, 32x32 protos)
    a128 = R.anchors_for(128, [int(128 / 544 * s) for s in (24, 48, 96, 192, 384)])
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=3, bg_bias=5.0)
    _nms_case(ref_out, 'small128', cls, box, coef, proto, a128, [(96, 128), (128, 64)], True)
    # (4) nothing above the score threshold -> five Nones
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=4, bg_bias=30.0)
    _nms_case(ref_out, 'empty128', cls, box, coef, proto, a128, [], True)
    # (5) fewer candidates than top_k, plus degenerate boxes: huge negative offsets push several
    #     boxes fully outside [0,1] -> clipped to zero area -> IoU 0/0 = NaN path
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=5, bg_bias=7.5)
    box[0, ::3, 0] = -40.0
    box[0, ::3, 2] = -8.0
    _nms_case(ref_out, 'degenerate128', cls, box, coef, proto, a128, [(64, 64)], True)
    # (6) exact score ties between identical anchors rows (duplicated predictions)
    cls, box, coef, proto = R.synth_head_outputs(1023, proto_hw=32, seed=6, bg_bias=5.0)
    cls[0, 1::2] = cls[0, 0::2][: cls[0, 1::2].shape[0]]
    _nms_case(ref_out, 'ties128', cls, box, coef, proto, a128, [(64, 64)], True)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_config, ref_yolact, ref_out, ref_box = import_reference()
    torch.set_num_threads(8)
    gen_state(ref_config, ref_yolact)
    gen_forward(ref_config, ref_yolact)
    gen_post(ref_config, ref_yolact, ref_out)
    print('goldens written to', OUT)


if __name__ == '__main__':
    main()
