"""Training goldens: the REAL reference's four losses + gradient digests on synthetic targets, and pins
oracle/yolact_ref.py's train-mode restatement (TrainNet + compute_loss) against it.  TEST INFRASTRUCTURE ONLY.
Run from the repo root: python oracle/make_golden_train.py"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import yolact_ref as R  # noqa: E402
from oracle.make_golden import import_reference, ref_cfg, tensor_digest, OUT  # noqa: E402


def main():
    ref_config, ref_yolact, ref_out, ref_box = import_reference()
    torch.set_num_threads(8)
    for name, size, batch, seed in (('res50_coco', 64, 2, 41), ('res50_coco', 128, 2, 42)):
        cfg = ref_cfg(ref_config, name, size, mode='train')
        torch.manual_seed(seed)
        net = ref_yolact.Yolact(cfg).train()
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        img = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(seed + 300))
        boxes, masks = R.synth_targets(batch, size, seed=seed)
        losses = net(img, [b.clone() for b in boxes], [m.clone() for m in masks])
        total = sum(losses)
        total.backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters()}
        sd1 = net.state_dict()

        # restatement on identical leaves
        params = {k: v.clone() for k, v in sd0.items()}
        for k, p in net.named_parameters():
            params[k].requires_grad_(True)
        tn = R.TrainNet(params)
        out = tn.forward(img)
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
        np.savez_compressed(
            os.path.join(OUT, f'train_{name}_{size}_b{batch}.npz'), seed=np.array(seed),
            losses=np.array([float(l) for l in losses], dtype=np.float64),
            grad_keys=np.array(keys), grad_digest=np.stack([tensor_digest(grads[k]) for k in keys]),
            grad_conv1=grads['backbone.conv1.weight'].numpy(),
            grad_bbox=grads['prediction_layers.bbox_layer.weight'].numpy()[:, :8],
            run_mean_stem=sd1['backbone.bn1.running_mean'].numpy(), run_var_stem=sd1['backbone.bn1.running_var'].numpy(),
            n_pos=np.array(0))
        print(name, size, 'losses', [round(float(l), 5) for l in losses], 'ok')


if __name__ == '__main__':
    main()
