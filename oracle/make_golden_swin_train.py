"""Swin-T TRAINING goldens from the REAL reference (swin_tiny_coco, train mode, DropPath switched off because its per-sample
torch.rand mask cannot be reproduced across devices) + pin of oracle/yolact_ref.py's `forward_train_any` + `compute_loss`:
losses and EVERY parameter gradient bit-equal.  TEST INFRASTRUCTURE ONLY.  Run: python oracle/make_golden_swin_train.py"""
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


if __name__ == '__main__':
    main()
