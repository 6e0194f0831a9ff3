#!/usr/bin/env python3
"""Where does a bs=1 conv launch spend its time OUTSIDE the workgroup bodies?  Runs a dependent chain of ResNet bottleneck convs
(conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual, layer3 shapes by default) with the trace build (`make -C yolact_minimal_amd/csrc
trace`): every launch stamps s_memtime per workgroup at entry / after the prologue / after the K loop / after the epilogue into
its own region.  Every XCD has its own counter base, so launches are compared inside one XCD: for consecutive launches the gap
`first entry of launch k+1  -  last exit of launch k` is the dead time of a dependent kernel boundary as the shader sees it.

    python tools/chain_trace.py [bs] [hw] [blocks]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['YM_LIB_PATH'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
from yolact_minimal_amd import hip  # noqa: E402
from tools.conv_sweep import make_desc  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 34
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wide, mid = {136: (256, 64), 68: (512, 128), 34: (1024, 256), 17: (2048, 512)}[hw]
dev = torch.device('cuda:0')
REGION = 4096 * 4
ws = torch.empty(1 << 27, dtype=torch.uint8, device=dev)
counters = torch.zeros(hip.TILE_COUNTERS, dtype=torch.int32, device=dev)
tuned = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolact_minimal_amd', 'tuned_gfx950.json')))

specs = [(bs, hw, hw, wide, mid, 1, 1, 0), (bs, hw, hw, mid, mid, 3, 1, 0), (bs, hw, hw, mid, wide, 1, 1, 1)]
descs = []
x_wide = torch.randn(bs, hw, hw, wide, device=dev)
for spec in specs:
    d, keep = make_desc(*spec, dev)
    sig = f'M{spec[0] * d.Ho * d.Wo}_N{spec[4]}_C{spec[3]}_k{spec[5]}_s{spec[6]}_seg1_r{spec[7]}'
    hit = tuned.get(sig, [0, 0, 0, 0, 0, 0, 0])
    d.tile_m, d.tile_n, d.ksplit, d.kwaves, d.stages = hit[0], hit[1], hit[2], hit[3], hit[4]
    d.tail_tiles, d.tail_ksplit = (hit[5], hit[6]) if len(hit) > 6 else (0, 0)
    d.tile_counters = counters.data_ptr()
    descs.append((sig, hit, d, keep))
# chain the buffers: conv1 reads the block input, conv2 reads conv1's output, conv3 reads conv2's output and writes the block input
descs[0][2].inp = x_wide.data_ptr()
descs[1][2].inp = descs[0][3][4].data_ptr()
descs[2][2].inp = descs[1][3][4].data_ptr()
descs[2][2].seg[0].out = x_wide.data_ptr()
descs[2][2].scale = None          # keep the values bounded over the chain: out = relu(0 * acc + shift + residual) stays O(1)
for sig, hit, d, _ in descs:
    print(sig, hit)

n = 3 * nblocks
trace = torch.zeros(n * REGION, dtype=torch.int64, device=dev)


def run(stamp):
    for k in range(n):
        if stamp:
            os.environ['YM_TRACE_PTR'] = str(trace.data_ptr() + k * REGION * 8)
        else:
            os.environ.pop('YM_TRACE_PTR', None)
        hip.conv2d_fwd(descs[k % 3][2], ws)


for _ in range(3):
    run(False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(False); e1.record()
torch.cuda.synchronize()
print(f'chain of {n} launches, no stamps: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch (eager, host enqueue may bound this)')
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        run(False)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    g.replay()
e1.record()
torch.cuda.synchronize()
print(f'chain of {n} launches, hipGraph replay: {e0.elapsed_time(e1) * 1e2 / n:.2f} us per launch')

trace.zero_()
e0.record(); run(True); e1.record()
torch.cuda.synchronize()
print(f'chain with stamps: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch')
raw = trace.cpu().reshape(n, REGION // 4, 4)
xcc = (raw[:, :, 0] >> 60) & 15
raw = raw.clone()
raw[:, :, 0] &= (1 << 60) - 1
raw = raw.double()
CLK = 2.38e3          # shader clock cycles per us (tools/micro/mfma_chain)
for k in range(n):
    ok = raw[k, :, 0] > 0
    agree = int(((xcc[k][ok] == (torch.arange(REGION // 4)[ok] % 8)).sum()))
    print(f'launch {k}: {int(ok.sum())} workgroups, block b on XCC b % 8 for {agree} of them; XCC of blocks 0..15: {xcc[k][:16].tolist()}')
for x in range(8):
    rows = []
    for k in range(n):
        r = raw[k][(xcc[k] == x) & (raw[k, :, 0] > 0) & (raw[k, :, 3] > 0)]
        if not r.shape[0]:
            rows.append(None)
            continue
        rows.append((float(r[:, 0].min()), float(r[:, 0].max()), float(r[:, 3].min()), float(r[:, 3].max()),
                     float((r[:, 1] - r[:, 0]).mean()), float((r[:, 2] - r[:, 1]).mean()), float((r[:, 3] - r[:, 2]).mean()), r.shape[0],
                     float((r[:, 3] - r[:, 0]).max())))
    if x not in (0, 5):
        continue
    print(f'XCC {x}: per launch [us]: WGs | entry spread | first entry -> last exit | longest WG | mean prologue / K loop / epilogue | gap to next launch')
    for k in range(n):
        if rows[k] is None:
            continue
        s0, s1, e0_, e1_, pro, kl, epi, cnt, longest = rows[k]
        gap = (rows[k + 1][0] - e1_) / CLK if k + 1 < n and rows[k + 1] is not None else float('nan')
        print(f'   {k:2d} {descs[k % 3][0]:34s} {cnt:4d} | {(s1 - s0) / CLK:5.2f} | {(e1_ - s0) / CLK:6.2f} | {longest / CLK:6.2f} | {pro / CLK:5.2f} / {kl / CLK:5.2f} / {epi / CLK:5.2f} '
              f'| {gap:6.2f}')
