"""Swin-T goldens from the REAL reference (swin_tiny_coco) + pin of oracle/yolact_ref.py's Swin restatement.
TEST INFRASTRUCTURE ONLY.  Run from the repo root: python oracle/make_golden_swin.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, ref_cfg, tensor_digest, OUT  # noqa: E402


def randomize_swin_(sd, seed):
    """LayerNorm affine / biases / relative-position tables are identity-like or zero at init: make them non-trivial."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if not k.startswith('backbone.'):
            continue
        if 'norm' in k and k.endswith('.weight'):
            sd[k].copy_(torch.rand(sd[k].shape, generator=g) * 0.4 + 0.8)
        elif k.endswith('.bias'):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.05)
        elif k.endswith('relative_position_bias_table'):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.5)
    return sd


def main():
    ref_config, ref_yolact, ref_out, ref_box = import_reference()
    torch.set_num_threads(8)
    # seeded construction digest
    cfg = ref_cfg(ref_config, 'swin_tiny_coco', 64)
    torch.manual_seed(7)
    net = ref_yolact.Yolact(cfg)
    sd = net.state_dict()
    keys = list(sd.keys())
    np.savez_compressed(os.path.join(OUT, 'state_swin.npz'), keys=np.array(keys), seed=np.array(7),
                        digest=np.stack([tensor_digest(sd[k].float()) for k in keys]))
    for size, batch, seed in ((128, 2, 51), (544, 1, 52)):
        cfg = ref_cfg(ref_config, 'swin_tiny_coco', size)
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).eval()
        sd = net.state_dict()
        randomize_swin_(sd, seed + 100)
        R.randomize_bias_(sd, seed + 200)
        net.load_state_dict(sd)
        img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
        with torch.no_grad():
            ref = net(img)
            mine = R.forward_eval_any(img, sd)
        for a, b in zip(ref, mine):
            assert torch.equal(a, b), 'swin restatement differs from the reference'
        if size == 128:
            np.savez_compressed(os.path.join(OUT, f'forward_swin_tiny_coco_{size}_b{batch}.npz'), seed=np.array(seed),
                                class_pred=ref[0].numpy(), box_pred=ref[1].numpy(), coef_pred=ref[2].numpy(),
                                proto_out=ref[3].numpy(), img_digest=tensor_digest(img))
        else:
            np.savez_compressed(os.path.join(OUT, 'forward_swin_tiny_coco_544_digest.npz'), seed=np.array(seed),
                                class_digest=tensor_digest(ref[0]), box_digest=tensor_digest(ref[1]),
                                coef_digest=tensor_digest(ref[2]), proto_digest=tensor_digest(ref[3]),
                                class_sample=ref[0][0, ::37].numpy(), box_sample=ref[1][0, ::37].numpy(),
                                coef_sample=ref[2][0, ::37].numpy(), proto_sample=ref[3][0, ::5, ::5].numpy(),
                                img_digest=tensor_digest(img))
        print('swin', size, 'ok; max class', float(ref[0].max()))


if __name__ == '__main__':
    main()
