"""Goldens of the reference's TRAINING LOOP (not one step): the REAL reference `Yolact` (imported from /root/reference) driven by
the statements of its own `train.py:60-63,102-130` on the CPU — `optim.SGD(net.parameters(), lr, momentum=0.9, weight_decay=5e-4)`
(AdamW(weight_decay=0.05) for swin_tiny_coco), warm-up / decay of `param_group['lr']`, `net(images, targets, masks)`,
`loss_total.backward()`, `optimizer.step()` — for a few steps on one synthetic batch, followed by the `evaluate`-style eval
forward of `train.py:165-166` on the trained weights.

  loop_<cfg>_<size>_b<B>.npz   per-step losses (fp32 CPU run), per-step learning rate, digests + strided samples of every parameter
                               AFTER the last step and of the parameter UPDATE (last - initial), stem running statistics and
                               `num_batches_tracked`, digests of the four eval outputs on image 0.

TEST INFRASTRUCTURE ONLY.  Run from the repo root:  python -m oracle.make_golden_loop [small] [swin] [full]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.optim as optim

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, tensor_digest, OUT  # noqa: E402


def sample(t):
    f = t.reshape(-1)
    return f[:: max(1, f.numel() // 64)][:64].clone()


def run(ref_config, ref_yolact, name, size, batch, seed, steps, damp=False):
    import argparse
    a = argparse.Namespace(cfg=name, img_size=size, weight=None, traditional_nms=False, val_num=-1, coco_api=False, resume=None,
                           train_bs=batch, bs_per_gpu=batch, val_interval=4000)       # train.py --train_bs = the batch (one process)
    a.mode, a.cuda, a.gpu_id = 'train', False, None
    cfg = getattr(ref_config, name)(a)
    torch.manual_seed(seed)
    net = ref_yolact.Yolact(cfg)
    net.train()
    if damp:
        # well-conditioned weights (near-identity residual blocks, ReLU zero crossings at -3 sigma): on the plain random init two
        # fp32 implementations differ by discrete ReLU sign flips, and a 3-step loop feeds that back into the weights (measured: the
        # HIP loop and this CPU loop, equal at step 0 to 6e-5, are 10 % apart in the mask loss one step later at 128 px bs=2)
        sd = net.state_dict()
        R.damp_residual_branches_(sd, seed + 400)
        R.shift_bn_bias_(sd, 3.0)                 # ... and no ReLU sign flips inside the rounding noise (see its docstring)
        net.load_state_dict(sd)
    p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    if 'res' in cfg.__class__.__name__:
        optimizer = optim.SGD(net.parameters(), lr=cfg.lr, momentum=0.9, weight_decay=5e-4)
    elif cfg.__class__.__name__ == 'swin_tiny_coco':
        optimizer = optim.AdamW(net.parameters(), lr=cfg.lr, weight_decay=0.05)
    img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
    boxes, masks = R.synth_targets(batch, size, seed=seed)
    if name.startswith('swin'):
        # DropPath draws from the global generator: the reference run is only reproducible from the same stream on the same
        # device, so the loop golden for Swin-T is taken with the stochastic depth switched off (every DropPath prob = 0)
        for m in net.modules():
            if m.__class__.__name__ == 'DropPath':
                m.drop_prob = 0.0
    losses, lrs, t0 = [], [], time.time()
    step = 0
    for _ in range(steps):
        images, targets, gt_masks = img, [b.clone() for b in boxes], [m.clone() for m in masks]
        if cfg.warmup_until > 0 and step <= cfg.warmup_until:
            for param_group in optimizer.param_groups:
                param_group['lr'] = (cfg.lr - cfg.warmup_init) * (step / cfg.warmup_until) + cfg.warmup_init
        if step in cfg.lr_steps:
            for param_group in optimizer.param_groups:
                param_group['lr'] = cfg.lr * 0.1 ** cfg.lr_steps.index(step)
        loss_c, loss_b, loss_m, loss_s = net(images, targets, gt_masks)
        loss_total = loss_c + loss_b + loss_m + loss_s
        optimizer.zero_grad()
        loss_total.backward()
        optimizer.step()
        losses.append([float(l) for l in (loss_c, loss_b, loss_m, loss_s)])
        lrs.append(optimizer.param_groups[0]['lr'])
        step += 1
        print(f'  step {step - 1}: lr {lrs[-1]:.3e} losses {[round(v, 5) for v in losses[-1]]} ({time.time() - t0:.1f}s)', flush=True)
    p1 = {k: v.detach().clone() for k, v in net.named_parameters()}
    sd1 = net.state_dict()
    net.eval()
    with torch.no_grad():
        out = net(img[:1])
    keys = list(p1.keys())
    stem = 'backbone.bn1' if 'backbone.bn1.running_mean' in sd1 else None
    np.savez_compressed(
        os.path.join(OUT, f'loop_{name}_{size}_b{batch}.npz'), seed=np.array(seed), steps=np.array(steps), damped=np.array(int(damp)),
        losses=np.array(losses, dtype=np.float64), lrs=np.array(lrs, dtype=np.float64), keys=np.array(keys),
        param_digest=np.stack([tensor_digest(p1[k]) for k in keys]),
        update_digest=np.stack([tensor_digest(p1[k] - p0[k]) for k in keys]),
        update_sample=np.stack([np.pad(sample(p1[k] - p0[k]).double().numpy(), (0, 64 - min(64, sample(p1[k]).numel()))) for k in keys]),
        update_absmax=np.array([float((p1[k] - p0[k]).abs().max()) for k in keys]),
        run_mean_stem=sd1[f'{stem}.running_mean'].numpy() if stem else np.zeros(0),
        run_var_stem=sd1[f'{stem}.running_var'].numpy() if stem else np.zeros(0),
        num_batches_tracked=np.array(int(sd1[f'{stem}.num_batches_tracked']) if stem else -1),
        eval_digest=np.stack([tensor_digest(o) for o in out]),
        eval_class_sample=out[0][0, ::37].numpy(), eval_proto_sample=out[3][0, ::5, ::5].numpy())
    print(f'loop {name}@{size} b{batch}: {steps} steps ok ({time.time() - t0:.1f}s)', flush=True)


def main():
    what = set(sys.argv[1:]) or {'small', 'swin'}
    ref_config, ref_yolact, _, _ = import_reference()
    torch.set_num_threads(8)
    if 'small' in what:
        run(ref_config, ref_yolact, 'res50_coco', 256, 4, 71, 3, damp=True)
    if 'swin' in what:
        run(ref_config, ref_yolact, 'swin_tiny_coco', 128, 2, 73, 3)
    if 'full' in what:
        run(ref_config, ref_yolact, 'res101_coco', 544, 8, 72, 3, damp=True)


if __name__ == '__main__':
    main()
