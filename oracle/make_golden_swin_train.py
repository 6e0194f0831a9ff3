"""Swin-T TRAINING goldens from the REAL reference (swin_tiny_coco, train mode, DropPath switched off because its per-sample
torch.rand mask cannot be reproduced across devices) + pin of oracle/yolact_ref.py's `forward_train_any` + `compute_loss`:
losses and EVERY parameter gradient bit-equal.  TEST INFRASTRUCTURE ONLY.  Run: python oracle/make_golden_swin_train.py [full]
(`full` adds the 544 px bs=8 step with fp64 gradient samples; a few minutes on 8 cores)"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, ref_cfg, tensor_digest, OUT  # noqa: E402
from oracle.make_golden_swin import randomize_swin_  # noqa: E402


def main():
    ref_config, ref_yolact, ref_out, ref_box = import_reference()
    torch.set_num_threads(8)
    for size, batch, seed in ((128, 2, 51),):
        cfg = ref_cfg(ref_config, 'swin_tiny_coco', size, mode='train')
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).train()
        for m in net.modules():                              # DropPath -> identity (drop_prob 0)
            if m.__class__.__name__ == 'DropPath':
                m.drop_prob = 0.
        with torch.no_grad():
            randomize_swin_(net.state_dict(), seed + 1)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
        boxes, masks = R.synth_targets(batch, size, seed=seed)
        losses = net(img, [b.clone() for b in boxes], [m.clone() for m in masks])
        sum(losses).backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}

        params = {k: v.clone() for k, v in sd0.items()}
        for k, _ in net.named_parameters():
            params[k].requires_grad_(True)
        out = R.forward_train_any(img, params)
        anchors = torch.tensor(net.anchors if isinstance(net.anchors, list) else net.anchors.tolist()).reshape(-1, 4)
        mine = R.compute_loss(*out, boxes, masks, anchors)
        for a, b in zip(losses, mine):
            assert torch.equal(a.detach(), b.detach()), (a, b)
        sum(mine).backward()
        for k in grads:
            assert torch.equal(grads[k], params[k].grad), k
        keys = list(grads.keys())
        np.savez_compressed(
            os.path.join(OUT, f'train_swin_tiny_coco_{size}_b{batch}.npz'), seed=np.array(seed),
            losses=np.array([float(l) for l in losses], dtype=np.float64),
            grad_keys=np.array(keys), grad_digest=np.stack([tensor_digest(grads[k]) for k in keys]),
            grad_table=grads['backbone.layers.0.blocks.1.attn.relative_position_bias_table'].numpy(),
            grad_qkv_bias=grads['backbone.layers.2.blocks.1.attn.qkv.bias'].numpy(),
            grad_patch_embed=grads['backbone.patch_embed.proj.weight'].numpy())
        print('swin_tiny_coco', size, 'losses', [round(float(l), 5) for l in losses], 'ok', len(keys), 'gradients bit-equal')
    if 'full' in sys.argv[1:]:
        full_size(ref_config, ref_yolact)


def full_size(ref_config, ref_yolact, size=544, batch=8, seed=52):
    """BASELINE config 5's backbone at the benchmarked size in TRAIN mode (bench: extra.train_swin_tiny_coco): the reference's step,
    the restatement pinned bit for bit at this size, and an fp64 evaluation of the same step — stored as strided fp64 samples of
    every gradient plus the reference run's own distance from fp64 per tensor (the yardstick of the GPU test), exactly like
    oracle/make_golden_fullsize.py does for the ResNets."""
    import time
    from oracle.make_golden_fullsize import grad_sample
    cfg = ref_cfg(ref_config, 'swin_tiny_coco', size, mode='train')
    torch.manual_seed(seed)
    net = ref_yolact.Yolact(cfg).train()
    for m in net.modules():
        if m.__class__.__name__ == 'DropPath':
            m.drop_prob = 0.
    with torch.no_grad():
        randomize_swin_(net.state_dict(), seed + 1)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
    boxes, masks = R.synth_targets(batch, size, seed=seed)
    t0 = time.time()
    losses = net(img, [b.clone() for b in boxes], [m.clone() for m in masks])
    sum(losses).backward()
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    print(f'reference step {time.time() - t0:.1f}s', flush=True)
    keys = list(grads.keys())
    anchors = torch.tensor(net.anchors if isinstance(net.anchors, list) else net.anchors.tolist()).reshape(-1, 4)

    params = {k: v.clone() for k, v in sd0.items()}
    for k in keys:
        params[k].requires_grad_(True)
    mine = R.compute_loss(*R.forward_train_any(img, params), boxes, masks, anchors)
    for a, b in zip(losses, mine):
        assert torch.equal(a.detach(), b.detach()), (a, b)
    sum(mine).backward()
    for k in keys:
        assert torch.equal(grads[k], params[k].grad), k

    t0 = time.time()
    p64 = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k in keys:
        p64[k].requires_grad_(True)
    o64 = R.forward_train_any(img.double(), p64)
    torch.set_default_dtype(torch.float64)
    try:
        l64 = R.compute_loss(*o64, [b.double() for b in boxes], [m.double() for m in masks], anchors.double())
    finally:
        torch.set_default_dtype(torch.float32)
    sum(l64).backward()
    e = np.array([((grads[k].double() - p64[k].grad).abs().max() / (p64[k].grad.abs().max() + 1e-30)).item() for k in keys])
    print(f'fp64 step {time.time() - t0:.1f}s; fp32-vs-fp64 gradient error / max|g|: median {np.median(e):.2e} p90 {np.quantile(e, 0.9):.2e} '
          f'max {e.max():.2e} ({keys[int(e.argmax())]})', flush=True)

    def samples(get):
        return np.stack([np.pad(grad_sample(get(k)).numpy(), (0, 64 - min(64, grad_sample(get(k)).numel()))) for k in keys])
    np.savez_compressed(
        os.path.join(OUT, f'train_swin_tiny_coco_{size}_b{batch}.npz'), seed=np.array(seed),
        losses=np.array([float(l.detach()) for l in losses], dtype=np.float64),
        losses_fp64=np.array([float(l.detach()) for l in l64], dtype=np.float64), grad_keys=np.array(keys),
        grad_digest=np.stack([tensor_digest(grads[k]) for k in keys]), grad_sample=samples(lambda k: grads[k]),
        grad_sample_fp64=samples(lambda k: p64[k].grad), grad_absmax=np.array([float(p64[k].grad.abs().max()) for k in keys]),
        grad_err_vs_fp64=e)
    print('swin_tiny_coco', size, batch, 'losses', [round(float(l), 5) for l in losses], 'restatement bit-equal: ok', flush=True)


if __name__ == '__main__':
    main()
