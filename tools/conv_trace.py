#!/usr/bin/env python3
"""In-kernel phase timing of conv_igemm_f32 (debug build tools/trace/libyolact_hip_trace.so): per workgroup
s_memtime stamps at kernel entry, after the prologue (first tile staged), after the K loop, after the epilogue."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

dev = torch.device('cuda:0')
trace = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
os.environ['YM_TRACE_PTR'] = str(trace.data_ptr())
ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
for spec, tile, ks in (((1, 34, 34, 256, 1024, 1, 1, 1), (64, 64), 1), ((1, 34, 34, 1024, 256, 1, 1, 0), (64, 64), 3),
                       ((1, 34, 34, 256, 256, 3, 1, 0), (64, 64), 6), ((8, 34, 34, 256, 256, 3, 1, 0), (64, 64), 1),
                       ((8, 136, 136, 256, 256, 3, 1, 0), (128, 128), 1)):
    d, keep = make_desc(*spec, dev)
    d.tile_m, d.tile_n, d.ksplit = tile[0], tile[1], ks
    for _ in range(3):
        hip.conv2d_fwd(d, ws)
    torch.cuda.synchronize()
    trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); hip.conv2d_fwd(d, ws); e1.record()
    torch.cuda.synchronize()
    t = trace.cpu().reshape(-1, 4)
    t = t[t[:, 0] > 0].double()
    M = spec[0] * d.Ho * d.Wo
    start0 = t[:, 0].min()
    print(f'M={M} N={spec[4]} K={d.k_pad} tile={tile} ks={ks}: {t.shape[0]} WGs, event {e0.elapsed_time(e1) * 1e3:.1f} us')
    print(f'   (shader clock cycles) WG start spread {(t[:, 0].max() - start0):.0f}; prologue {((t[:, 1] - t[:, 0]).mean()):.0f}; '
          f'K loop {((t[:, 2] - t[:, 1]).mean()):.0f}; epilogue {((t[:, 3] - t[:, 2]).mean()):.0f}; '
          f'first start -> last end {(t[:, 3].max() - start0):.0f}')
