"""Goldens at the BENCHMARKED sizes, from the REAL reference (imported from /root/reference in the build container):

  forward_<cfg>_544_b8_digest.npz   eval forward of res50_coco / res101_coco / swin_tiny_coco at 544 px, batch 8
                                    (BASELINE.json configs 2 and 5, and config 3's per-GPU shape): per-image digests
                                    (sum, |sum|, sum of squares) + strided samples of all four outputs
  train_res50_coco_256_b4.npz       well-conditioned training step (layer4's BatchNorms see 256 samples): 4 losses + digests
                                    and strided samples of every parameter gradient + stem running stats
  train_res101_coco_544_b8.npz      config 3's per-GPU training step at full size (same content)
  train_res101_coco_544_b16.npz     config 4's per-GPU training step (batch 16): 4 losses + fp32 gradient digests
  train_res101_coco_544_b8_wellcond.npz config 3's shape, residual branches damped (R.damp_residual_branches_) and the backbone's ReLU
                                    zero crossings moved to -3 sigma (R.shift_bn_bias_: no sign flips inside the rounding noise): the fp32
                                    reference sits within ~1e-5 of fp64, the HIP step is held to 1e-3 of max|g| per tensor / 1 % on the digests

Every case also pins oracle/yolact_ref.py's restatement bit for bit against the reference.  TEST INFRASTRUCTURE ONLY.
Run from the repo root:  python -m oracle.make_golden_fullsize [forward] [train256] [train544] [train544res50] [train544b16] [train544wellcond]
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, ref_cfg, tensor_digest, OUT  # noqa: E402
from oracle.make_golden_swin import randomize_swin_  # noqa: E402


def per_image_digest(t):
    return np.stack([tensor_digest(t[b]) for b in range(t.shape[0])])


def gen_forward(ref_config, ref_yolact):
    for name, seed in (('res50_coco', 61), ('res101_coco', 62), ('swin_tiny_coco', 63)):
        cfg = ref_cfg(ref_config, name, 544)
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).eval()
        sd = net.state_dict()
        if name.startswith('swin'):
            randomize_swin_(sd, seed + 100)
        else:
            R.randomize_bn_(sd, seed + 100)
        R.randomize_bias_(sd, seed + 200)
        net.load_state_dict(sd)
        img = torch.randn(8, 3, 544, 544, generator=torch.Generator().manual_seed(seed + 300))
        t0 = time.time()
        with torch.no_grad():
            ref = net(img)
            mine = R.forward_eval_any(img, sd)
        for a, b in zip(ref, mine):
            assert torch.equal(a, b), f'oracle restatement differs from the reference ({name}@544 b8)'
        np.savez_compressed(
            os.path.join(OUT, f'forward_{name}_544_b8_digest.npz'), seed=np.array(seed),
            class_digest=per_image_digest(ref[0]), box_digest=per_image_digest(ref[1]),
            coef_digest=per_image_digest(ref[2]), proto_digest=per_image_digest(ref[3]),
            class_sample=ref[0][:, ::97].numpy(), box_sample=ref[1][:, ::97].numpy(),
            coef_sample=ref[2][:, ::97].numpy(), proto_sample=ref[3][:, ::9, ::9].numpy(),
            img_digest=tensor_digest(img))
        print(f'forward {name}@544 b8 ok ({time.time() - t0:.1f}s); max class {float(ref[0].max()):.4f}', flush=True)


def grad_sample(g):
    f = g.reshape(-1)
    return f[:: max(1, f.numel() // 64)][:64].clone()


def gen_train(ref_config, ref_yolact, name, size, batch, seed, damp=False, fp64=True, tag='', shift=0.0):
    cfg = ref_cfg(ref_config, name, size, mode='train')
    torch.manual_seed(seed)
    net = ref_yolact.Yolact(cfg).train()
    if damp:                                  # well-conditioned variant: near-identity residual blocks (see the docstring there)
        sd = net.state_dict()
        R.damp_residual_branches_(sd, seed + 400)
        if shift:
            R.shift_bn_bias_(sd, shift)              # ReLU zero crossings at -shift sigma: no sign flips inside the rounding noise
        net.load_state_dict(sd)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
    boxes, masks = R.synth_targets(batch, size, seed=seed)
    t0 = time.time()
    losses = net(img, [b.clone() for b in boxes], [m.clone() for m in masks])
    sum(losses).backward()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    sd1 = net.state_dict()
    print(f'reference step {time.time() - t0:.1f}s', flush=True)

    params = {k: v.clone() for k, v in sd0.items()}
    for k, p in net.named_parameters():
        params[k].requires_grad_(True)
    out = R.TrainNet(params).forward(img)
    anchors = torch.tensor(net.anchors if isinstance(net.anchors, list) else net.anchors.tolist()).reshape(-1, 4)
    mine = R.compute_loss(*out, boxes, masks, anchors)
    for a, b in zip(losses, mine):
        assert torch.equal(a.detach(), b.detach()), (a, b)
    sum(mine).backward()
    for k in grads:
        assert torch.equal(grads[k], params[k].grad), k
    for k in sd1:
        if 'running' in k:
            assert torch.equal(sd1[k], params[k].detach()), k
    keys = list(grads.keys())
    if not fp64:          # (bs=16 at 544 px: the fp64 evaluation would take ~10 min on 8 cores; losses + fp32 digests only)
        np.savez_compressed(
            os.path.join(OUT, f'train_{name}_{size}_b{batch}.npz'), seed=np.array(seed),
            losses=np.array([float(l.detach()) for l in losses], dtype=np.float64), grad_keys=np.array(keys),
            grad_digest=np.stack([tensor_digest(grads[k]) for k in keys]),
            run_mean_stem=sd1['backbone.bn1.running_mean'].numpy(), run_var_stem=sd1['backbone.bn1.running_var'].numpy())
        print(name, size, batch, 'losses', [round(float(l), 5) for l in losses], 'restatement bit-equal: ok (no fp64 pass)', flush=True)
        return
    # how far is this fp32 run from an fp64 evaluation of the same step?  (per tensor, relative to max|g|: the tests' yardstick)
    t0 = time.time()
    p64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k in keys:
        p64[k].requires_grad_(True)
    o64 = R.TrainNet(p64).forward(img.double())
    torch.set_default_dtype(torch.float64)
    try:
        l64 = R.compute_loss(*o64, [b.double() for b in boxes], [m.double() for m in masks], anchors.double())
    finally:
        torch.set_default_dtype(torch.float32)
    sum(l64).backward()
    e = np.array([((grads[k].double() - p64[k].grad).abs().max() / (p64[k].grad.abs().max() + 1e-30)).item() for k in keys])
    print(f'fp64 step {time.time() - t0:.1f}s; fp32-vs-fp64 gradient error / max|g|: median {np.median(e):.2e} '
          f'p90 {np.quantile(e, 0.9):.2e} max {e.max():.2e} ({keys[int(e.argmax())]}); '
          f'loss rel err {max(abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(losses, l64)):.2e}', flush=True)
    np.savez_compressed(
        os.path.join(OUT, f'train_{name}_{size}_b{batch}{tag}.npz'), seed=np.array(seed),
        losses=np.array([float(l.detach()) for l in losses], dtype=np.float64),
        losses_fp64=np.array([float(l.detach()) for l in l64], dtype=np.float64),
        grad_keys=np.array(keys), grad_digest=np.stack([tensor_digest(grads[k]) for k in keys]),
        grad_sample=np.stack([np.pad(grad_sample(grads[k]).numpy(), (0, 64 - min(64, grad_sample(grads[k]).numel()))) for k in keys]),
        grad_sample_fp64=np.stack([np.pad(grad_sample(p64[k].grad).numpy(), (0, 64 - min(64, grad_sample(p64[k].grad).numel()))) for k in keys]),
        grad_absmax=np.array([float(p64[k].grad.abs().max()) for k in keys]), grad_err_vs_fp64=e,
        run_mean_stem=sd1['backbone.bn1.running_mean'].numpy(), run_var_stem=sd1['backbone.bn1.running_var'].numpy())
    print(name, size, batch, 'losses', [round(float(l), 5) for l in losses], 'restatement bit-equal: ok', flush=True)


def main():
    what = set(sys.argv[1:]) or {'forward', 'train256', 'train544', 'train544res50', 'train544b16'}
    ref_config, ref_yolact, ref_out, ref_box = import_reference()
    torch.set_num_threads(8)
    if 'forward' in what:
        gen_forward(ref_config, ref_yolact)
    if 'train256' in what:
        gen_train(ref_config, ref_yolact, 'res50_coco', 256, 4, 71, damp=True)
    if 'train544' in what:
        gen_train(ref_config, ref_yolact, 'res101_coco', 544, 8, 72)
    if 'train544res50' in what:       # the second ResNet depth at the benchmarked size (bench: extra.res50_coco_bs8 / CPU baseline config 1)
        gen_train(ref_config, ref_yolact, 'res50_coco', 544, 8, 74)
    if 'train544wellcond' in what:    # the WELL-CONDITIONED full-size step: config 3's shape, near-identity residual blocks, ReLU crossings at -3 sigma
        gen_train(ref_config, ref_yolact, 'res101_coco', 544, 8, 75, damp=True, shift=3.0, tag='_wellcond')
    if 'train544b16' in what:         # BASELINE config 4's per-GPU batch
        gen_train(ref_config, ref_yolact, 'res101_coco', 544, 16, 73, fp64=False)


if __name__ == '__main__':
    main()
