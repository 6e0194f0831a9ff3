"""Train -> evaluate -> mAP on a tiny synthetic dataset, all on the HIP path: the closest offline stand-in for the north star's
acceptance line ("box / mask mAP within 0.1 of the reference on COCO val2017": no dataset and no trained weights here).

What it shows: the training step (`Trainer.step`: forward, the four losses, backward, SGD with the reference's schedule,
train.py:102-130) drives a RANDOM-INIT res50 YOLACT to a working detector on images it has seen, and the evaluation path
(`net.eval()` forward -> `nms` -> `after_nms` -> `prep_metrics` -> `calc_map`, eval.py:36-110) reports it — i.e. the two halves
of the path agree about boxes, classes, prototypes and coefficients beyond the per-step parity the goldens pin.

Data: `--images` pictures of `--size` px with 2-3 filled shapes each (class = shape and colour: rectangle / ellipse / triangle /
diamond, the four `CUSTOM_CLASSES` slots of `res50_custom`), on a noisy background; boxes are the shapes' tight boxes, masks their
pixels.  Everything is seeded.

    python tools/overfit_demo.py --steps 1500            # prints the loss every 100 steps and the final mAP table (JSON last)
    YM_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/overfit_demo.py --size 128 \
        --steps 600                                      # two ranks of 4 pictures each (sharing one GPU), rank 0 scores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

COLOURS = np.array([[1.6, -0.8, -0.8], [-0.8, 1.6, -0.8], [-0.8, -0.8, 1.6], [1.4, 1.4, -1.0]], np.float32)


def shape_mask(kind, size, cx, cy, rx, ry):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    dx, dy = (xx - cx) / rx, (yy - cy) / ry
    if kind == 0:                                   # rectangle
        return (np.abs(dx) <= 1) & (np.abs(dy) <= 1)
    if kind == 1:                                   # ellipse
        return dx * dx + dy * dy <= 1
    if kind == 2:                                   # upright triangle (apex on top)
        return (dy <= 1) & (dy >= -1) & (np.abs(dx) <= (dy + 1) / 2)
    return np.abs(dx) + np.abs(dy) <= 1             # diamond


def make_dataset(n_images, size, seed=0):
    """[(image [3, S, S] float32 (already normalised), gt [n, 5] (x1 y1 x2 y2 in 0..1, class), masks [n, S, S] float32)]."""
    rng = np.random.default_rng(seed)
    data = []
    for _ in range(n_images):
        img = rng.normal(0.0, 0.25, (3, size, size)).astype(np.float32)
        taken = np.zeros((size, size), bool)
        gts, masks = [], []
        for _ in range(int(rng.integers(2, 4))):
            for _try in range(20):
                kind = int(rng.integers(0, 4))
                rx, ry = rng.uniform(0.09, 0.2, 2) * size
                cx, cy = rng.uniform(rx + 2, size - rx - 2), rng.uniform(ry + 2, size - ry - 2)
                m = shape_mask(kind, size, cx, cy, rx, ry)
                if m.sum() > 40 and not (m & taken).any():
                    break
            else:
                continue
            taken |= m
            ys, xs = np.nonzero(m)
            gts.append([xs.min() / size, ys.min() / size, (xs.max() + 1) / size, (ys.max() + 1) / size, kind])
            masks.append(m.astype(np.float32))
            img[:, m] = COLOURS[kind][:, None] + rng.normal(0.0, 0.1, (3, int(m.sum()))).astype(np.float32)
        data.append((torch.from_numpy(img), torch.tensor(gts, dtype=torch.float32), torch.from_numpy(np.stack(masks))))
    return data


def evaluate(net, cfg, data, device, size):
    """eval.py:36-110 on the training pictures: one image at a time, fast-NMS, masks at the picture's own size."""
    from yolact_minimal_amd.utils.common_utils import APDataObject, prep_metrics, calc_map
    from yolact_minimal_amd.utils.output_utils import nms, after_nms
    thres = [x / 100 for x in range(50, 100, 5)]
    nc = len(cfg.class_names)
    ap = {k: [[APDataObject() for _ in range(nc)] for _ in thres] for k in ('box', 'mask')}
    net.eval()
    found = 0
    strong = [0]                      # detections above 0.3 (what a user would draw)
    with torch.no_grad():
        for img, gt, masks in data:
            out = net(img[None].to(device))
            ids, sc, bx, cf, pr = nms(out[0], out[1], out[2], out[3], net.anchors, cfg)
            ids, sc, boxes_p, masks_p = after_nms(ids, sc, bx, cf, pr, size, size, cfg)
            if ids is None:
                continue
            found += 1
            strong[0] += int((sc > 0.3).sum())
            prep_metrics(ap, list(ids.cpu().numpy().astype(int)), list(sc.cpu().numpy().astype(float)), boxes_p, masks_p,
                         gt.clone().to(device), masks.to(device), size, size, thres)
    table, row_box, row_mask = calc_map(ap, thres, nc, step=0)
    net.train()
    evaluate.detections_above_0p3 = strong[0]
    return table, row_box, row_mask, found


def serving_agrees(net, cfg, data, device, size, depth=4):
    """The trained detector through the serving path (`RequestPipeline`: `depth` requests in flight, hipGraph engines, batched nms /
    after_nms kernels, one pinned count read per request) against eval.py's sequential calls, picture by picture: how many pictures
    come back with identical ids / scores / pixel boxes / masks."""
    from yolact_minimal_amd.pipeline import RequestPipeline
    from yolact_minimal_amd.utils.output_utils import nms, after_nms
    net.eval()
    same, n_det = 0, 0
    with torch.no_grad():
        imgs = [d[0][None].to(device).contiguous() for d in data]
        seq = []
        for x in imgs:
            out = net(x)
            seq.append(after_nms(*nms(out[0], out[1], out[2], out[3], net.anchors, cfg), size, size, cfg))
        pipe = RequestPipeline(net, cfg, size, size, device, depth=depth, out_hw=(size, size))
        pipe.warm_up(imgs[0], rounds=1)
        res = []
        for x in imgs:
            r = pipe.submit(x)
            if r is not None:
                res.append(r)
        res += [r for r in pipe.drain() if r is not None]
    assert len(res) == len(seq), (len(res), len(seq))
    for a, b in zip(seq, res):
        if a[0] is None or b[0] is None:
            same += a[0] is None and b[0] is None
            continue
        n_det += int(a[0].shape[0])
        same += all(x.shape == y.shape and bool(torch.equal(x, y)) for x, y in zip(a, b))
    net.train()
    return same, n_det


def run(steps=1500, n_images=16, size=256, batch=8, cfg_name='res50_custom', seed=0, lr=None, log=print, eval_every=0, log_every=100):
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    from yolact_minimal_amd.trainer import Trainer, init_distributed, shard_batch
    # under torch.distributed.run: `batch` is the GLOBAL batch (train.py --train_bs), every rank trains its contiguous shard of the
    # same seeded pick, rank 0 scores (YM_DIST_BACKEND=gloo lets the ranks share one GPU; RCCL refuses duplicate devices)
    rank, world, local_rank = init_distributed()
    device = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    mine = shard_batch(batch, rank, world)
    cfg = build_cfg(cfg_name, 'train', size, train_bs=batch, bs_per_gpu=batch // world)
    if lr is not None:
        cfg.lr = lr
    torch.manual_seed(seed)
    net = Yolact(cfg)
    tr = Trainer(net, cfg, device, world, local_rank % torch.cuda.device_count())
    if rank != 0:
        log = lambda *_: None                                                   # noqa: E731
    data = make_dataset(n_images, size, seed)
    imgs = torch.stack([d[0] for d in data]).to(device)
    gts = [d[1].to(device) for d in data]
    mks = [d[2].to(device) for d in data]
    order = np.random.default_rng(seed + 1)
    hist, curve = [], []
    t0 = time.time()
    for step in range(steps):
        pick = order.choice(n_images, batch, replace=False)[list(mine)]
        losses = tr.step(imgs[pick], [gts[i] for i in pick], [mks[i] for i in pick])
        if step % log_every == 0 or step == steps - 1:
            vals = [round(float(l.detach()), 4) for l in losses]
            hist.append((step, vals))
            log(f'step {step:5d}  lr {tr.opt.lr:.5f}  loss c/b/m/s {vals}  total {sum(vals):.3f}')
        if eval_every and step and step % eval_every == 0:
            _, rb, rm, _ = evaluate(net, cfg, data, device, size)
            curve.append((step, rb[1], rm[1]))
            log(f'         box mAP {rb[1]}  mask mAP {rm[1]}')
    torch.cuda.synchronize()
    train_s = time.time() - t0
    replicas_identical = None
    if world > 1:
        import torch.distributed as dist
        digest = torch.stack([tr.opt.flat.double().sum(), tr.opt.flat.double().abs().sum(), tr.opt.buf.double().abs().sum()])
        digest = digest if dist.get_backend() == 'nccl' else digest.cpu()
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(gathered, digest)
        replicas_identical = all(bool(torch.equal(gathered[0].cpu(), g.cpu())) for g in gathered)
        if rank != 0:
            dist.barrier()                       # (rank 0 scores alone, like train.py:162-174)
            return None
    table, row_box, row_mask, found = evaluate(net, cfg, data, device, size)
    strong = evaluate.detections_above_0p3
    # the same detector through `--traditional_nms` (greedy per-class NMS, utils/output_utils.py:84-123 + cython_nms.pyx)
    cfg.traditional_nms = True
    _, trad_box, trad_mask, _ = evaluate(net, cfg, data, device, size)
    cfg.traditional_nms = False
    same, n_det = serving_agrees(net, cfg, data, device, size)
    if world > 1:
        dist.barrier()
    return dict(cfg=cfg_name, size=size, images=n_images, batch=batch, steps=steps, train_s=round(train_s, 1), losses=hist,
                box_map=row_box[1:], mask_map=row_mask[1:], images_with_detections=found, curve=curve, table=table,
                serving_path_identical_pictures=same, detections=n_det, detections_above_0p3=strong, world=world, replicas_identical=replicas_identical,
                box_map_traditional_nms=trad_box[1:], mask_map_traditional_nms=trad_mask[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=1500)
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--cfg', default='res50_custom')
    ap.add_argument('--lr', type=float, default=None)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--eval-every', type=int, default=0)
    ap.add_argument('--log-every', type=int, default=100)
    a = ap.parse_args()
    r = run(a.steps, a.images, a.size, a.batch, a.cfg, seed=a.seed, lr=a.lr, eval_every=a.eval_every, log_every=a.log_every)
    if r is not None:                            # (rank 0)
        print(r.pop('table'))
        print('OVERFIT ' + json.dumps(r) if r['world'] > 1 else json.dumps(r))


if __name__ == '__main__':
    main()
