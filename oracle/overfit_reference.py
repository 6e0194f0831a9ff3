"""The REAL reference (imported from /root/reference, CPU) trained by the statements of its own `train.py:60-63,102-130` on the
synthetic shapes dataset of `tools/overfit_demo.py` — same seeded initial weights, same pictures, same batch order, same
schedule — so that the LONG-RUN behaviour of the HIP training path (hundreds of steps: does the loss fall, do the weights stay
finite, what mAP do the training pictures reach) has the reference's own run beside it.  Trajectories of two fp32
implementations separate (discrete ReLU / OHEM / positive-selection flips feed back into the weights), so what is compared is
the loss level per phase and the final mAP, not tensors.

TEST INFRASTRUCTURE ONLY (this container only: needs /root/reference).
    python -m oracle.overfit_reference --size 128 --steps 600 --out tests/golden/overfit_reference_128.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.optim as optim

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))
from oracle import yolact_ref as R  # noqa: E402
from oracle import metrics_ref as M  # noqa: E402
from oracle.make_golden import import_reference  # noqa: E402
from overfit_demo import make_dataset  # noqa: E402  (a pure input generator)


def evaluate(net, data, size, nc, anchors):
    thres = [x / 100 for x in range(50, 100, 5)]
    ap = M.new_ap_data(nc, len(thres))
    net.eval()
    found = 0
    n_det = [0, 0]                    # detections returned, detections above 0.3
    with torch.no_grad():
        for img, gt, masks in data:
            out = net(img[None])
            r = R.nms(out[0], out[1], out[2], out[3], anchors, img_size=size)
            if r[0] is None:
                continue
            ids, sc, boxes_p, masks_p = R.after_nms(r[0], r[1], r[2], r[3], r[4], size, size)
            if ids is None:
                continue
            found += 1
            n_det[0] += len(ids)
            n_det[1] += int((sc > 0.3).sum())
            M.prep_metrics(ap, [int(i) for i in ids], [float(s) for s in sc], boxes_p, masks_p, gt.clone(), masks, size, size, thres)
    net.train()
    res = M.calc_map(ap, thres, nc)
    return res, found, n_det


def main():
    a = argparse.ArgumentParser()
    a.add_argument('--size', type=int, default=128)
    a.add_argument('--steps', type=int, default=600)
    a.add_argument('--images', type=int, default=16)
    a.add_argument('--batch', type=int, default=8)
    a.add_argument('--cfg', default='res50_custom')
    a.add_argument('--seed', type=int, default=0)
    a.add_argument('--lr', type=float, default=None)
    a.add_argument('--log-every', type=int, default=10)
    a.add_argument('--out', default='')
    args = a.parse_args()
    ref_config, ref_yolact, _, _ = import_reference()
    # under torch.distributed.run (gloo, CPU): train.py:76's DDP(net) over the ranks, `--batch` = train.py --train_bs (global), every
    # rank trains its contiguous shard of the same seeded pick (DistributedSampler-style), rank 0 scores
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    torch.set_num_threads(max(1, 8 // world))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend='gloo', init_method='env://')
    per = args.batch // world
    ns = argparse.Namespace(cfg=args.cfg, img_size=args.size, weight=None, traditional_nms=False, val_num=-1, coco_api=False,
                            resume=None, train_bs=args.batch, bs_per_gpu=args.batch // world, val_interval=4000)
    ns.mode, ns.cuda, ns.gpu_id = 'train', False, None
    cfg = getattr(ref_config, args.cfg)(ns)
    if args.lr is not None:
        cfg.lr = args.lr
    torch.manual_seed(args.seed)
    net = ref_yolact.Yolact(cfg)
    net.train()
    module = net
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(net, broadcast_buffers=True)       # train.py:76 (CPU: no device ids)
    if 'res' in cfg.__class__.__name__:                                   # train.py:60-63
        optimizer = optim.SGD(net.parameters(), lr=cfg.lr, momentum=0.9, weight_decay=5e-4)
    else:
        optimizer = optim.AdamW(net.parameters(), lr=cfg.lr, weight_decay=0.05)
    data = make_dataset(args.images, args.size, args.seed)
    imgs = torch.stack([d[0] for d in data])
    order = np.random.default_rng(args.seed + 1)
    hist, t0 = [], time.time()
    for step in range(args.steps):
        pick = order.choice(args.images, args.batch, replace=False)[rank * per:(rank + 1) * per]
        if cfg.warmup_until > 0 and step <= cfg.warmup_until:
            for g in optimizer.param_groups:
                g['lr'] = (cfg.lr - cfg.warmup_init) * (step / cfg.warmup_until) + cfg.warmup_init
        if step in cfg.lr_steps:
            for g in optimizer.param_groups:
                g['lr'] = cfg.lr * 0.1 ** cfg.lr_steps.index(step)
        losses = net(imgs[pick], [data[i][1].clone() for i in pick], [data[i][2].clone() for i in pick])
        total = losses[0] + losses[1] + losses[2] + losses[3]
        optimizer.zero_grad()
        total.backward()
        optimizer.step()
        if rank == 0 and (step % args.log_every == 0 or step == args.steps - 1):
            vals = [round(float(l), 4) for l in losses]
            hist.append((step, vals))
            print(f'step {step:5d}  lr {optimizer.param_groups[0]["lr"]:.5f}  loss c/b/m/s {vals}  ({time.time() - t0:.0f}s)', flush=True)
    if world > 1 and rank != 0:
        dist.barrier()
        return
    net = module
    anchors = torch.tensor(net.anchors).reshape(-1, 4)
    res, found, n_det = evaluate(net, data, args.size, len(cfg.class_names), anchors)
    out = dict(cfg=args.cfg, size=args.size, images=args.images, batch=args.batch, steps=args.steps, seed=args.seed, lr=cfg.lr,
               losses=hist, box_map=[round(v, 2) for v in res['box']], mask_map=[round(v, 2) for v in res['mask']],
               images_with_detections=found, detections=n_det[0], detections_above_0p3=n_det[1], cpu_s=round(time.time() - t0, 1), world=world)
    print(json.dumps(out))
    if args.out:
        json.dump(out, open(args.out, 'w'))
    if world > 1:
        dist.barrier()


if __name__ == '__main__':
    main()
