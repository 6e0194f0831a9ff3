"""CHAINED end-to-end golden from the REAL reference (eval.py:45-52): res101_coco 544 px bs=1,

    class_p, box_p, coef_p, proto_p = net(img)                       (modules/yolact.py:141-164)
    ids, class_p, box_p, coef_p, proto_p = nms(..., net.anchors, cfg)  (utils/output_utils.py:126-163)
    ids, class_p, boxes, masks = after_nms(..., 480, 640, cfg)        (utils/output_utils.py:200-233)

where nms / after_nms consume the forward's OWN outputs (every other post-processing golden feeds synthetic head outputs: a
random-init network gives degenerate detections).  To get a few hundred real candidates out of seeded random weights the
prediction head's conf-layer weight is scaled up and its bias re-centred per (anchor, class), see `shape_head_`; the seeds are
searched until the detection SET (compared as `oracle.yolact_ref.detections_match` does: partners of equal class with score and
box within 1e-4; neighbouring anchors of a random-init network have scores 1e-7 apart, so the ORDER among them is not a property
of the network) is unchanged under +-2e-6 noise on the scores / boxes (20 draws): no suppression decision and no top-k cut flips.
Also pins oracle/yolact_ref.py bit for bit on this chain.  TEST INFRASTRUCTURE ONLY.

Run from the repo root:  python -m oracle.make_golden_chained
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, ref_cfg, tensor_digest, OUT  # noqa: E402


CONF_GAIN = 10.0
TARGET_CANDIDATES = 400


def shape_head_(sd, img, seed, conf_gain=CONF_GAIN, bias_std=1.0):
    """In place: make the shared prediction head produce confident, SPATIALLY VARYING class scores from random features.  The conf
    logits of a random-init net are a per-(anchor, class) constant (std 0.26 over classes) plus a small spatial variation (std
    0.07-0.2): the conf weight is scaled by `conf_gain` and the bias is set to MINUS the spatial mean of every (anchor, class)
    logit plus N(0, bias_std), plus a background offset found by bisection such that ~TARGET_CANDIDATES anchors pass the 0.05
    score threshold.  Returns the bias (stored in the golden: the test rebuilds the network from the seed, this gain and this
    bias)."""
    g = torch.Generator().manual_seed(seed)
    w = sd['prediction_layers.conf_layer.weight']
    b = sd['prediction_layers.conf_layer.bias']
    w.mul_(conf_gain)
    b.zero_()
    conf = R.features_any(img, sd)[0]                       # [1, N, 81] logits with a zero bias
    la = conf[0].reshape(-1, 3, 81)                         # anchor index = (y*W + x)*3 + a  ->  channel a*81 + c
    nb = torch.randn(3, 81, generator=g) * bias_std - la.mean(dim=0)

    def candidates(bg):
        bb = nb.clone()
        bb[:, 0] += bg
        p = torch.softmax((la + bb).reshape(-1, 81), -1)
        return int((p[:, 1:].max(dim=1)[0] > 0.05).sum())
    lo, hi = 0.0, 40.0
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if candidates(mid) > TARGET_CANDIDATES:
            lo = mid
        else:
            hi = mid
    nb[:, 0] += round(hi, 3)
    b.copy_(nb.reshape(-1))
    return b.clone()


def build(ref_config, ref_yolact, seed):
    cfg = ref_cfg(ref_config, 'res101_coco', 544)
    torch.manual_seed(seed)
    net = ref_yolact.Yolact(cfg).eval()
    sd = net.state_dict()
    R.randomize_bn_(sd, seed + 100)
    R.randomize_bias_(sd, seed + 200)
    img = torch.randn(1, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300))
    bias = shape_head_(sd, img, seed + 500)
    net.load_state_dict(sd)
    return cfg, net, sd, img, bias


def stable(out, anchors, ref, trials=20, eps=2e-6):
    """Is the oracle's detection set unchanged -- as `R.detections_match` compares sets, i.e. up to swaps among scores closer than
    1e-4 -- when the network outputs move by ~eps (what a different summation order does)?  Neighbouring anchors of a random-init
    network have nearly equal scores, so an exact-order criterion cannot be met; suppression decisions (IoU against 0.5) and the
    per-class top-k cut must not flip."""
    cls, box, coef, proto = out
    g = torch.Generator().manual_seed(7)
    for _ in range(trials):
        c2 = cls + (torch.rand(cls.shape, generator=g) * 2 - 1) * eps * cls.abs().clamp_min(1e-3)
        b2 = box + (torch.rand(box.shape, generator=g) * 2 - 1) * eps * box.abs().clamp_min(1e-2)
        r = R.nms(c2, b2, coef, proto, anchors)
        ok, msg, pairs = R.detections_match(ref, r)
        if not ok or len(pairs) < ref[0].numel() - 3:
            return False
    return True


def main():
    ref_config, ref_yolact, ref_out, _ = import_reference()
    for seed in range(71, 90):
        cfg, net, sd, img, conf_bias = build(ref_config, ref_yolact, seed)
        with torch.no_grad():
            out = net(img)
            mine = R.forward_eval_any(img, sd)
        for a, b in zip(out, mine):
            assert torch.equal(a, b), 'oracle forward restatement differs from the reference'
        anchors_list = net.anchors
        anchors = torch.tensor(anchors_list, dtype=torch.float32).reshape(-1, 4)
        with torch.no_grad():
            r = ref_out.nms(out[0].clone(), out[1].clone(), out[2].clone(), out[3].clone(), anchors_list, cfg)
            m = R.nms(out[0], out[1], out[2], out[3], anchors)
        if r[0] is None:
            print(f'seed {seed}: no detections'); continue
        n_cand = int((out[0][0, :, 1:].max(dim=1)[0] > cfg.nms_score_thre).sum())
        for a, b in zip(r[:4], m[:4]):
            assert torch.equal(a, b), 'oracle nms restatement differs from the reference on the chained outputs'
        n = int(r[0].numel())
        sc = r[1]
        gap = float((sc[:-1] - sc[1:]).min()) if n > 1 else 1.0
        ok = n >= 20 and len(set(r[0].tolist())) >= 3 and stable(out, anchors, r)
        print(f'seed {seed}: {n_cand} candidates over the threshold, {n} detections, {len(set(r[0].tolist()))} classes, '
              f'min score gap {gap:.2e}, stable={ok}', flush=True)
        if not ok:
            continue
        with torch.no_grad():
            ra = ref_out.after_nms(r[0], r[1], r[2].clone(), r[3], r[4], 480, 640)       # (eval.py:51: no cfg)
            ma = R.after_nms(m[0], m[1], m[2], m[3], m[4], 480, 640)
        assert torch.equal(ra[2], ma[2]) and torch.equal(ra[3], ma[3]), 'oracle after_nms restatement differs'
        np.savez_compressed(
            os.path.join(OUT, 'chained_res101_coco_544.npz'), seed=np.array(seed), n=np.array(n), candidates=np.array(n_cand),
            class_digest=tensor_digest(out[0]), box_digest=tensor_digest(out[1]), coef_digest=tensor_digest(out[2]),
            proto_digest=tensor_digest(out[3]), ids=r[0].numpy(), scores=r[1].numpy(), boxes=r[2].numpy(), coefs=r[3].numpy(),
            px_boxes=ra[2].numpy(), masks_packed=np.packbits(ra[3].numpy().astype(np.uint8).reshape(-1)),
            masks_area=ra[3].sum(dim=(1, 2)).numpy(), min_score_gap=np.array(gap), conf_gain=np.array(CONF_GAIN),
            conf_bias=conf_bias.numpy())
        print(f'chained_res101_coco_544.npz written (seed {seed})')
        return
    raise SystemExit('no seed gave a well-separated detection set')


if __name__ == '__main__':
    main()
