#!/usr/bin/env python3
"""Requests in flight on DISJOINT shares of the CUs (cu_mask.py) instead of competing for all of them: forward-only throughput of a
RequestPipeline with `parts` slots, (a) as shipped (plain streams, the table's rows), (b) masked streams with the same rows,
(c) masked streams with rows measured on one share (InferEngine.autotune(cus=256 // parts) under the masked stream).
  python tools/cu_partition_probe.py [--parts 4] [--out rows.json]"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolact_minimal_amd import engine as E, pipeline as P  # noqa: E402
from tools import cu_mask  # noqa: E402
from yolact_minimal_amd.engine import InferEngine  # noqa: E402


def throughput(pipe, img, n=240):
    pipe.warm_up(img)
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.submit(img)
        pipe.drain()
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0))
    return round(best, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--parts', type=int, default=4)
    ap.add_argument('--cfg', default='res101_coco')
    ap.add_argument('--size', type=int, default=544)
    ap.add_argument('--out', default='')
    ap.add_argument('--post', action='store_true', help='forward + nms + after_nms per request (the headline workload) instead of the forward alone')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    net, cfg = bench.build_net(args.cfg, args.size, dev)
    img = torch.randn(1, 3, args.size, args.size, device=dev)
    kw = dict(depth=args.parts, with_post=args.post, return_outputs=False)
    out = dict(parts=args.parts, cus_per_share=256 // args.parts, cfg=args.cfg, size=args.size)
    head = None
    if args.post:
        from yolact_minimal_amd.utils.synthetic import synth_head_outputs
        head = [t.to(dev) for t in synth_head_outputs(len(net.anchors) // 4, num_classes=cfg.num_classes, proto_hw=args.size // 4, seed=1)]

    def run(tag):
        pipe = P.RequestPipeline(net, cfg, args.size, args.size, dev, **kw)
        if head is not None:
            sub = pipe.submit
            pipe.submit = lambda im, h=None: sub(im, head)
        out[tag] = throughput(pipe, img)
        print(tag, out[tag], flush=True)
        del pipe
        torch.cuda.empty_cache()
    # hipExtStreamCreateWithCUMask makes BLOCKING streams (no flags argument): anything on the legacy null stream -- torch's default
    # current stream, where submit() records the event a slot waits for -- joins all of them.  So the caller works on a stream of
    # its own and the null stream stays empty.
    caller = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(caller)
    run('plain_streams_table_rows')
    streams = [cu_mask.masked_stream(dev, i, args.parts) for i in range(args.parts)]
    P._streams[torch.device(dev)] = streams
    run('masked_streams_table_rows')
    eng = InferEngine(net, 1, args.size, args.size, dev, mode='latency')
    with torch.cuda.stream(streams[0]):
        rows = eng.autotune(iters=10, cus=256 // args.parts)
    torch.cuda.synchronize()
    rows = {k: (v if len(v) > 7 and v[7] else v[:7]) for k, v in rows.items()}
    table = E.tuned_table()
    for k in rows:
        table.pop(k + '_tp', None)
    table.update(rows)
    del eng
    run('masked_streams_share_rows')
    if args.out:
        json.dump(rows, open(args.out, 'w'), indent=0, sort_keys=True)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
