#!/usr/bin/env python3
"""Phase stamps of the single-workgroup nms kernels (debug build: `make -C yolact_minimal_amd/csrc trace`, then
YM_LIB_PATH=tools/trace/libyolact_hip_trace.so python tools/nms_stamps.py): s_memrealtime (100 MHz) at the phase boundaries of
k_class_topk_iou (class 0) and of k_final_select (stage C),
on the bench's dense synthetic head outputs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from yolact_minimal_amd.utils import output_utils as OU  # noqa: E402
from yolact_minimal_amd.utils.synthetic import synth_head_outputs  # noqa: E402

dev = torch.device('cuda:0')
net, cfg = bench.build_net('res101_coco', 544, dev)
head = [t.to(dev) for t in synth_head_outputs(len(net.anchors) // 4, num_classes=cfg.num_classes, proto_hw=136, seed=1)]
anchors = torch.tensor(net.anchors, dtype=torch.float32).reshape(-1, 4).to(dev)
for it in range(5):
    d = OU.nms_batch(*head, anchors, cfg)
    torch.cuda.synchronize()
    ws = next(iter(OU._ws_cache.values()))
    st = ws[:256].view(torch.int32)[8:8 + 32].view(torch.int64).cpu().tolist()
    m, c, t = st[0:5], st[8:12], st[12:15]
    if it >= 2:
        print('k_class_topk_iou (class 0): select+sort %.2f us (key loads + radix passes %.2f, collect %.2f, sort %.2f), IoU '
              'columns %.2f us, compaction %.2f us' % ((c[1] - c[0]) / 100, (t[1] - c[0]) / 100, (t[2] - t[1]) / 100,
                                                       (c[1] - t[2]) / 100, (c[2] - c[1]) / 100, (c[3] - c[2]) / 100))
        print('stage C: counts %.2f us, key load %.2f us, select + rank %.2f us, gather %.2f us;  class 0 start -> stage C start %.2f us' %
              ((m[1] - m[0]) / 100, (m[2] - m[1]) / 100, (m[3] - m[2]) / 100, (m[4] - m[3]) / 100, (m[0] - c[0]) / 100))
