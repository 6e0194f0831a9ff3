#!/usr/bin/env python3
"""Per-shape timing of conv_wgrad_f32 (+ reduce) for every W_ entry of the tuned table (res101/res50 bs=8 training shapes)."""
import ctypes
import json
import math
import os
import re
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolact_minimal_amd import hip  # noqa: E402
from yolact_minimal_amd.hip import WgradDesc  # noqa: E402

dev = torch.device('cuda:0')
table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd', 'tuned_gfx950.json')))
IN_OF = {272: 544, 136: 136, 68: 68, 34: 34, 17: 17, 9: 9, 5: 5}
S2_IN = {272: 544, 68: 136, 34: 68, 17: 34, 9: 17, 5: 9}
big = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
rows = []
for key, hit in sorted(table.items()):
    m = re.match(r'W_M(\d+)_N(\d+)_C(\d+)_k(\d+)_s(\d+)$', key)
    if not m:
        continue
    M, n, c, k, s = (int(v) for v in m.groups())
    b = 8
    ho = int(round(math.sqrt(M / b)))
    if ho * ho * b != M or (s == 2 and ho not in S2_IN):
        continue                      # entries of other batch sizes / the pyramid-batched head
    h = S2_IN[ho] if s == 2 else ho
    pad = k // 2
    x = torch.randn(b, h, h, c, device=dev)
    dy = torch.randn(b, ho, ho, n, device=dev)
    cin_real = 3 if c == 4 else c
    dw = torch.empty(n, cin_real, k, k, device=dev)
    d = WgradDesc()
    d.x, d.dy, d.dw = x.data_ptr(), dy.data_ptr(), dw.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cin_real, d.Cout, d.Cout_real = b, h, h, c, cin_real, n, n
    d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo, d.msplit = k, k, s, pad, ho, ho, hit[0]
    d.lds_buffers = hit[1] if len(hit) > 1 else 2
    need = hip.lib().ym_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    if need == 0 or need > big.numel():
        print(key, 'skipped', need)
        continue

    def run():
        hip.check(hip.lib().ym_conv2d_wgrad(ctypes.byref(d), ctypes.c_void_p(big.data_ptr()), big.numel(), hip.stream_ptr()), 'wgrad')
    for _ in range(2):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    gf = 2.0 * M * n * k * k * c / 1e9
    rows.append((key, hit[0], us, gf / us * 1e-3 * 1e3))
    del x, dy, dw
for key, ms, us, tf in sorted(rows, key=lambda r: -r[2]):
    print(f'{key:34s} msplit={ms:4d} {us:9.1f} us {tf:7.1f} TF')
