#!/usr/bin/env python3
"""The reference-shaped eval loop (dropin/reference_loops.eval_loop, prep_metrics branch) without the timer's stage fences, with
host-side stamps per statement: where the host waits when nothing fences the stages (measured: in nms's count read, i.e. on the forward; HSA_ENABLE_INTERRUPT=0 /
ROC_ACTIVE_WAIT_TIMEOUT / a stream synchronize before the read change nothing: 3.5 ms per image with fences, 3.35 without).
  python tools/eval_loop_probe.py [--images 40]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=40)
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    from yolact_minimal_amd.utils.synthetic import synth_eval_case
    from yolact_minimal_amd.utils.output_utils import nms, after_nms
    from yolact_minimal_amd.utils.common_utils import prep_metrics, APDataObject
    L = bench._dropin_loops()
    net, cfg, img = bench.detecting_net('res101_coco', 544, dev)
    _, _, _, _, gt, gt_masks, _, _ = synth_eval_case(1, 40, 15, 480, 640, 10)
    gt, gt_masks = gt.to(dev), gt_masks.to(dev)
    out = dict(tag=args.tag)
    for fences in (True, False):
        L.eval_loop(net, cfg, [(img, gt.clone(), gt_masks, 480, 640) for _ in range(3)], sync_stages=fences)
        _, _, seen, secs = L.eval_loop(net, cfg, [(img, gt.clone(), gt_masks, 480, 640) for _ in range(args.images)], sync_stages=fences)
        out['fences' if fences else 'no_fences'] = round(secs / args.images * 1e3, 3)
    # host stamps per statement, no fences
    thres = [x / 100 for x in range(50, 100, 5)]
    acc = [0.0] * 4
    for i in range(args.images + 3):
        ap_data = {k: [[APDataObject() for _ in cfg.class_names] for _ in thres] for k in ('box', 'mask')}
        t0 = time.perf_counter()
        with torch.no_grad():
            o = net(img)
        t1 = time.perf_counter()
        r = nms(*o, net.anchors, cfg)
        t2 = time.perf_counter()
        ids_p, class_p, boxes_p, masks_p = after_nms(*r, 480, 640)
        t3 = time.perf_counter()
        prep_metrics(ap_data, list(ids_p.cpu().numpy().astype(int)), list(class_p.cpu().numpy().astype(float)), boxes_p, masks_p, gt.clone(),
                     gt_masks, 480, 640, thres)
        t4 = time.perf_counter()
        if i >= 3:
            for j, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                acc[j] += d
    torch.cuda.synchronize()
    out['host_ms'] = dict(zip(('forward_call', 'nms_call', 'after_nms_call', 'metric_call'), (round(a / args.images * 1e3, 3) for a in acc)))
    # the device-RLE metric branch, statement by statement (host stamps; every .cpu() is a synchronising D2H)
    from yolact_minimal_amd.utils.common_utils import rle_encode, MakeJson
    mj = MakeJson()
    names = ('ids_cpu', 'scores_cpu', 'boxes_cpu', 'rle_encode', 'records')
    acc = [0.0] * len(names)
    for i in range(args.images + 3):
        with torch.no_grad():
            o = net(img)
        ids_p, class_p, boxes_p, masks_p = after_nms(*nms(*o, net.anchors, cfg), 480, 640)
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        ids = list(ids_p.cpu().numpy().astype(int)); t.append(time.perf_counter())
        sc = list(class_p.cpu().numpy().astype(float)); t.append(time.perf_counter())
        bx = boxes_p.cpu().numpy(); t.append(time.perf_counter())
        rles = rle_encode(masks_p); t.append(time.perf_counter())
        for j in range(len(rles)):
            mj.add_bbox(i, ids[j], bx[j, :], sc[j])
            mj.add_mask(i, ids[j], rles[j], sc[j])
        t.append(time.perf_counter())
        if i >= 3:
            for j in range(len(names)):
                acc[j] += t[j + 1] - t[j]
    for rep in range(2):
        L.eval_loop(net, cfg, [(img, gt.clone(), gt_masks, 480, 640) for _ in range(3)], coco_api='device')
        _, _, seen, secs = L.eval_loop(net, cfg, [(img, gt.clone(), gt_masks, 480, 640) for _ in range(args.images)], coco_api='device')
        out[f'eval_loop_device_rle_{rep}'] = dict(ms=round(secs / args.images * 1e3, 3), metric_ms=round(L.timer.get_times(['metric'])[0] * 1e3, 3))
    out['rle_branch_host_ms'] = dict(zip(names, (round(a / args.images * 1e3, 3) for a in acc)))
    out['rle_string_bytes'] = sum(len(r['counts']) for r in rles)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
