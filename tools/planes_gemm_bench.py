#!/usr/bin/env python3
"""EXPERIMENT: fp32-grade GEMM on the bf16 MFMA with pre-split operand planes (tools/micro/planes_gemm.hip) against the f32 MFMA
conv kernel, on the 1x1 conv shapes of the bs=8 plan.  Reports time (GEMM alone, and + the split pass of the activation), TFLOP/s
(f32-equivalent) and max error vs fp64 for both."""
import ctypes
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, 'micro', 'libplanes_gemm.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-shared', '-fPIC',
                           os.path.join(HERE, 'micro', 'planes_gemm.hip'), '-o', so])
L = ctypes.CDLL(so)
L.pg_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
L.pg_gemm.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
dev = torch.device('cuda:0')
tuned = json.load(open(os.path.join(os.path.dirname(HERE), 'yolact_minimal_amd', 'tuned_gfx950.json')))
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (b, hw, cin, cout) in ((8, 34, 256, 1024), (8, 68, 512, 128), (8, 136, 2304, 256), (8, 136, 256, 256)):
    M, K, N = b * hw * hw, cin, cout
    d, keep = make_desc(b, hw, hw, cin, cout, 1, 1, 0, dev)
    x, wt, sc, sh, out, _ = keep
    sig = f'M{M}_N{N}_C{K}_k1_s1_seg1_r0'
    hit = tuned.get(sig) or [0, 0, 0, 0, 0, 0, 0]
    d.tile_counters = counters.data_ptr()
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
    d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.grid_wgs = hit[7] if len(hit) > 7 else 0
    t_f32 = timeit(lambda: hip.conv2d_fwd(d, ws))
    ref = torch.relu((x.reshape(M, K).double() @ wt.double().t()) * sc.double() + sh.double())
    err_f32 = float((out.reshape(M, N).double() - ref).abs().max() / ref.abs().max())
    pa = torch.empty(M * K * 3, dtype=torch.int16, device=dev)
    pw = torch.empty(N * K * 3, dtype=torch.int16, device=dev)
    L.pg_split(x.data_ptr(), pa.data_ptr(), M, K, st)
    L.pg_split(wt.data_ptr(), pw.data_ptr(), N, K, st)
    out2 = torch.empty(M, N, device=dev)
    flops = 2.0 * M * N * K
    line = f'M{M} N{N} K{K}: f32 MFMA {hit} {t_f32:7.1f} us {flops / t_f32 / 1e6:6.1f} TF err {err_f32:.1e} |'
    t_split = timeit(lambda: L.pg_split(x.data_ptr(), pa.data_ptr(), M, K, st))
    for ns, pf in ((3, 0), (6, 0), (4, 1), (6, 1), (4, 2), (6, 2)):       # pf 2 = k_gemm2: no vector instructions in the K loop
        def run():
            rc = L.pg_gemm(pa.data_ptr(), pw.data_ptr(), out2.data_ptr(), M, N, K, sc.data_ptr(), sh.data_ptr(), 1, ns, pf, st)
            assert rc == 0, rc
        out2.zero_()
        run()
        torch.cuda.synchronize()
        err = float((out2.double() - ref).abs().max() / ref.abs().max())
        t = timeit(run)
        line += f' planes ns{ns}{("", "pf", "v2")[pf]} {t:6.1f} us {flops / t / 1e6:6.1f} TF err {err:.1e} |'
    for ns in (3, 6):
        for abl, name in ((1, 'DMA only'), (2, 'MFMA + LDS reads only')):
            t = timeit(lambda: L.pg_gemm(pa.data_ptr(), pw.data_ptr(), out2.data_ptr(), M, N, K, sc.data_ptr(), sh.data_ptr(), 1 | (abl << 4), ns, 0, st))
            line += f' [ns{ns} {name}: {t:.1f} us]'
    print(line + f' split pass {t_split:.1f} us', flush=True)
