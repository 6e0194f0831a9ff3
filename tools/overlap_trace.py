#!/usr/bin/env python3
"""Do the requests of bench.py's headline mode really overlap on the chip?  Evidence that does not need a tracer: rocprofv3
serialises much of the overlap away (profiles/r03_infer_bs1_inflight4_res101_gaps.txt: 1.18 kernels in flight), so the conv
kernels stamp themselves.

Runs `RequestPipeline` (res101_coco 544 px, depth requests in flight, forward + nms + after_nms like bench.py) on the trace build of
the library (`make -C yolact_minimal_amd/csrc trace`): every conv_igemm_f32 workgroup writes s_memtime at entry / exit, its XCC id
and HW_REG_HW_ID (CU, shader engine, compute pipe, queue) into a per-(slot, launch) region; a device-side epoch word bumped by a
captured kernel at the head of every graph replay selects one of RING regions, so the last RING requests of every slot stay
readable.  The stamps are s_memrealtime (one constant-rate counter for the whole chip; the shader-clock counter s_memtime has a
different base on every CU group).  Per XCD (= per L2):

  * time-weighted histogram of how many REQUESTS (slots) / conv LAUNCHES have workgroups resident at the same instant,
  * resident workgroups and distinct busy CUs,
  * which compute pipe / queue each slot's workgroups came from,
inside the steady-state window (all slots busy), next to the wall-clock ms/step of the same run (trace build) and of the product
build.

    python tools/overlap_trace.py [--depth 4] [--batch 1] [--requests 96] [--out profiles/r04_infer_bs1_inflight4_overlap]
"""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument('--depth', type=int, default=4)
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--requests', type=int, default=96)
ap.add_argument('--ring', type=int, default=8)
ap.add_argument('--cfg', default='res101_coco')
ap.add_argument('--out', default='')
ap.add_argument('--product', action='store_true', help='(internal) time the same loop on the product build and print ms/step')
args = ap.parse_args()

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
if not args.product:
    os.environ['YM_LIB_PATH'] = os.path.join(REPO, 'tools', 'trace', 'libyolact_hip_trace.so')
import torch  # noqa: E402
import bench  # noqa: E402
from yolact_minimal_amd import hip, engine as E  # noqa: E402
from yolact_minimal_amd.pipeline import RequestPipeline  # noqa: E402
from yolact_minimal_amd.utils.synthetic import synth_head_outputs  # noqa: E402

dev = torch.device('cuda:0')
net, cfg = bench.build_net(args.cfg, 544, dev)
img = torch.randn(args.batch, 3, 544, 544, generator=torch.Generator().manual_seed(0)).to(dev)
head = [t.to(dev).expand(args.batch, *t.shape[1:]).contiguous()
        for t in synth_head_outputs(len(net.anchors) // 4, num_classes=cfg.num_classes, proto_hw=136, seed=1)]
RING = args.ring


def timed_loop(pipe, n):
    for _ in range(2 * args.depth):
        pipe.submit(img, head)
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit(img, head)
    pipe.drain()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if args.product:
    pipe = RequestPipeline(net, cfg, 544, 544, dev, depth=args.depth, out_hw=(480, 640), batch=args.batch, return_outputs=False)
    pipe.warm_up(img)
    print(json.dumps(dict(ms_per_step=round(min(timed_loop(pipe, args.requests) for _ in range(3)), 4))))
    sys.exit(0)

# ---- instrument: every conv launch of every slot gets its own stamp regions; the epoch word of a slot is bumped inside its graph ----
state = dict(eng=None, idx=0)
regions = {}          # (slot, conv index) -> (trace tensor [RING, grid, 4] int64, hw tensor [RING, grid] int32, grid, name)
MAXGRID = 8192
orig_conv = hip.conv2d_fwd
orig_launch_all = E.InferEngine._launch_all


def conv2d_fwd(desc, ws):
    eng = state['eng']
    if eng is not None:
        key = (eng._slot, state['idx'])
        ent = regions.get(key)
        if ent is None:
            ent = regions[key] = (torch.zeros(RING, MAXGRID, 4, dtype=torch.int64, device=dev),
                                  torch.zeros(RING, MAXGRID, dtype=torch.int32, device=dev))
        os.environ['YM_TRACE_PTR'] = str(ent[0].data_ptr())
        os.environ['YM_TRACE_HW_PTR'] = str(ent[1].data_ptr())
        os.environ['YM_TRACE_EPOCH_PTR'] = str(eng._epoch.data_ptr())
        os.environ['YM_TRACE_RING'] = str(RING)
        os.environ['YM_TRACE_GRID'] = str(MAXGRID)
        state['idx'] += 1
    return orig_conv(desc, ws)


def launch_all(self, x):
    if getattr(self, '_slot', None) is not None:
        self._epoch.add_(1)                      # captured: every replay of this slot's graph selects the next region
        state['eng'], state['idx'] = self, 0
    try:
        return orig_launch_all(self, x)
    finally:
        state['eng'] = None
        for k in ('YM_TRACE_PTR', 'YM_TRACE_HW_PTR', 'YM_TRACE_EPOCH_PTR', 'YM_TRACE_RING', 'YM_TRACE_GRID'):
            os.environ.pop(k, None)


hip.conv2d_fwd = conv2d_fwd
E.hip.conv2d_fwd = conv2d_fwd
E.InferEngine._launch_all = launch_all

pipe = RequestPipeline(net, cfg, 544, 544, dev, depth=args.depth, out_hw=(480, 640), batch=args.batch, return_outputs=False)
for s, e in enumerate(pipe.engines):
    e._slot = s
    e._epoch = torch.zeros(1, dtype=torch.int32, device=dev)
pipe.warm_up(img)
ms_trace = min(timed_loop(pipe, args.requests) for _ in range(2))
epochs = [int(e._epoch.item()) for e in pipe.engines]
names = [c.name for c in pipe.engines[0].convs]

# product build, same loop, in a child process (the library is chosen at import time)
import subprocess  # noqa: E402
env = dict(os.environ)
env.pop('YM_LIB_PATH', None)
child = subprocess.run([sys.executable, os.path.abspath(__file__), '--product', '--depth', str(args.depth), '--batch', str(args.batch),
                        '--requests', str(args.requests), '--cfg', args.cfg], env=env, capture_output=True, text=True)
ms_product = json.loads([l for l in child.stdout.splitlines() if l.startswith('{')][-1])['ms_per_step'] if child.returncode == 0 else None

# ---- read the stamps ----------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402
recs = []             # (xcc, t_in, t_out, slot, replay, conv, hw)
for (slot, ci), (tr, hw) in regions.items():
    tr, hw = tr.cpu().numpy(), hw.cpu().numpy()
    E_s = epochs[slot]
    for r in range(RING):
        # region r holds replay e with e % RING == r, the newest such e <= E_s
        e = E_s - ((E_s - r) % RING)
        blk = tr[r]
        live = blk[:, 0] != 0
        if not live.any():
            continue
        t0 = blk[live, 0]
        xcc = (t0.astype(np.uint64) >> np.uint64(60)).astype(np.int64)
        t_in = (t0.astype(np.uint64) & np.uint64((1 << 60) - 1)).astype(np.int64)
        t_out = blk[live, 3]
        ok = t_out > t_in
        for x, a, b, h in zip(xcc[ok], t_in[ok], t_out[ok], hw[r][live][ok]):
            recs.append((int(x), int(a), int(b), slot, int(e), ci, int(h)))
recs = np.array(recs, dtype=np.int64)
if os.environ.get('YM_OVERLAP_DUMP'):                # raw records of one XCD for offline inspection
    np.save(os.environ['YM_OVERLAP_DUMP'], recs[recs[:, 0] == 1])
covered = sorted({names[c] for c in set(recs[:, 5].tolist())})
summary = dict(workload=f'{args.cfg} 544x544 bs={args.batch}: forward + nms + after_nms(480x640), {args.depth} requests in flight, '
                        f'{args.requests} timed requests', ring=RING, stamped_workgroups=int(len(recs)),
               conv_launches_per_request=len(names), stamped_launches_per_request=len(set(recs[:, 5].tolist())),
               ms_per_step_trace_build=round(ms_trace, 4), ms_per_step_product_build=ms_product, epochs=epochs)

# sanity of the stamps: the conv launches of ONE request run on one stream, so on every XCD launch c+1 of a (slot, replay) must start
# after launch c ended (a violation means stamps of different replays were mixed)
order_checked = order_bad = 0
for x in range(8):
    Rx = recs[recs[:, 0] == x]
    for s in range(args.depth):
        Rs = Rx[Rx[:, 3] == s]
        for e in set(Rs[:, 4].tolist()):
            Q = Rs[Rs[:, 4] == e]
            cs = sorted(set(Q[:, 5].tolist()))
            spans = [(Q[Q[:, 5] == c][:, 1].min(), Q[Q[:, 5] == c][:, 2].max()) for c in cs]
            order_checked += max(0, len(spans) - 1)
            order_bad += sum(1 for (a0, b0), (a1, b1) in zip(spans, spans[1:]) if a1 < b0)
summary['launch_order_check'] = dict(consecutive_pairs=order_checked, overlapping_pairs=order_bad)

per_xcd = []
pipes_by_slot = {}
for x in range(8):
    R = recs[recs[:, 0] == x]
    if len(R) == 0:
        continue
    # steady window on this XCD's clock: every slot contributes its replays E-RING+2 .. E-2 (the newest ones ran with fewer in flight)
    lo, hi = [], []
    for s in range(args.depth):
        Rs = R[R[:, 3] == s]
        es = sorted(set(Rs[:, 4].tolist()))
        keep = [e for e in es if epochs[s] - RING + 2 <= e <= epochs[s] - 2]
        if not keep:
            continue
        lo.append(Rs[Rs[:, 4] == keep[0]][:, 1].min())
        hi.append(Rs[Rs[:, 4] == keep[-1]][:, 2].max())
    if len(lo) < args.depth:
        continue
    w0, w1 = max(lo), min(hi)
    if w1 <= w0:
        continue
    W = R[(R[:, 2] > w0) & (R[:, 1] < w1)]
    ev = []
    for i, (xc, a, b, s, e, c, h) in enumerate(W.tolist()):
        ev.append((max(a, w0), 1, i))
        ev.append((min(b, w1), 0, i))
    ev.sort()
    act_slot, act_launch, act_cu = {}, {}, {}
    n_wg = 0
    hist_req, hist_launch = {}, {}
    acc_wg = acc_cu = 0.0
    prev = w0
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            hist_req[len(act_slot)] = hist_req.get(len(act_slot), 0) + dt
            hist_launch[len(act_launch)] = hist_launch.get(len(act_launch), 0) + dt
            acc_wg += n_wg * dt
            acc_cu += len(act_cu) * dt
            prev = t
        _, a, b, s, e, c, h = W[i].tolist()
        cu = (h >> 8) & 0xFF                         # HW_ID: cu_id[11:8], sh_id[12], se_id[15:13]
        for d, k in ((act_slot, s), (act_launch, (s, e, c)), (act_cu, cu)):
            if kind == 1:
                d[k] = d.get(k, 0) + 1
            else:
                d[k] -= 1
                if d[k] == 0:
                    del d[k]
        n_wg += 1 if kind == 1 else -1
    span = float(w1 - w0)
    # request completions inside the window -> ms/step needs the tick rate: slot periods (first entry of consecutive replays) are
    # depth x ms_per_step in steady state
    periods = []
    for s in range(args.depth):
        Rs = R[R[:, 3] == s]
        starts = [Rs[Rs[:, 4] == e][:, 1].min() for e in sorted(set(Rs[:, 4].tolist())) if epochs[s] - RING + 2 <= e <= epochs[s] - 2]
        periods += [b - a for a, b in zip(starts, starts[1:])]
    ticks_per_request_period = float(np.median(periods)) if periods else None
    ghz = ticks_per_request_period / (args.depth * ms_trace * 1e6) if ticks_per_request_period else None     # counter ticks per ns
    per_xcd.append(dict(xcd=x, window_ticks=int(span), workgroups=int(len(W)),
                        requests_resident_hist={str(k): round(v / span, 4) for k, v in sorted(hist_req.items())},
                        mean_requests_resident=round(sum(k * v for k, v in hist_req.items()) / span, 3),
                        launches_resident_hist={str(k): round(v / span, 4) for k, v in sorted(hist_launch.items())},
                        mean_launches_resident=round(sum(k * v for k, v in hist_launch.items()) / span, 3),
                        mean_workgroups_resident=round(acc_wg / span, 2), mean_busy_cus=round(acc_cu / span, 2),
                        counter_ghz_implied=round(ghz, 3) if ghz else None))
    for s in range(args.depth):
        hs = R[R[:, 3] == s][:, 6]
        for h in np.unique(hs).tolist():
            key = f'me{(h >> 30) & 3}.pipe{(h >> 6) & 3}.queue{(h >> 24) & 7}'
            pipes_by_slot.setdefault(str(s), {}).setdefault(key, 0)
            pipes_by_slot[str(s)][key] += int((hs == h).sum())

summary['per_xcd'] = per_xcd
if per_xcd:
    summary['mean_requests_resident'] = round(float(np.mean([p['mean_requests_resident'] for p in per_xcd])), 3)
    summary['mean_launches_resident'] = round(float(np.mean([p['mean_launches_resident'] for p in per_xcd])), 3)
    summary['mean_busy_cus_per_xcd'] = round(float(np.mean([p['mean_busy_cus'] for p in per_xcd])), 2)
    summary['mean_workgroups_resident_chip'] = round(float(np.sum([p['mean_workgroups_resident'] for p in per_xcd])), 1)
summary['compute_pipe_of_each_slot'] = pipes_by_slot
summary['hw_id_samples'] = [hex(int(v) & 0xFFFFFFFF) for v in np.unique(recs[:, 6])[:12].tolist()]
summary['stamped_layers'] = covered[:6] + (['...'] if len(covered) > 6 else [])
summary['note'] = ('a launch counts as resident on an XCD while any of its stamped workgroups is between entry and exit there; only '
                   'conv_igemm_f32 launches stamp (conv_wave / non-conv kernels do not), so the true concurrency is at least this')
txt = [f"# overlap of {args.depth} requests in flight ({summary['workload']})",
       f"wall clock: {ms_trace:.3f} ms/step on the trace build, {ms_product} ms/step on the product build (same loop, same box)",
       f"stamped conv workgroups: {len(recs)} ({summary['stamped_launches_per_request']} of {len(names)} conv launches per request stamp); "
       f"stamp sanity: {order_bad} of {order_checked} consecutive launches of one request overlap on an XCD (must be 0)",
       '', '| XCD | requests resident: P(0) P(1) P(2) P(3) P(4+) | mean requests | mean launches | mean WGs | mean busy CUs (of 32) |', '|---|---|---|---|---|---|']
for p_ in per_xcd:
    h = p_['requests_resident_hist']
    p4 = sum(v for k, v in h.items() if int(k) >= 4)
    txt.append(f"| {p_['xcd']} | {h.get('0', 0):.3f} {h.get('1', 0):.3f} {h.get('2', 0):.3f} {h.get('3', 0):.3f} {p4:.3f} | "
               f"{p_['mean_requests_resident']} | {p_['mean_launches_resident']} | {p_['mean_workgroups_resident']} | {p_['mean_busy_cus']} |")
txt += ['', f"chip mean: {summary.get('mean_requests_resident')} requests / {summary.get('mean_launches_resident')} conv launches resident per XCD at "
            f"any instant, {summary.get('mean_busy_cus_per_xcd')} busy CUs per XCD, {summary.get('mean_workgroups_resident_chip')} conv workgroups on the chip",
        f"compute pipe / queue of each slot's workgroups: {json.dumps(pipes_by_slot)}", summary['note']]
print('\n'.join(txt))
print(json.dumps(summary))
if args.out:
    os.makedirs(os.path.dirname(os.path.join(REPO, args.out)) or '.', exist_ok=True)
    with open(os.path.join(REPO, args.out + '.json'), 'w') as f:
        json.dump(summary, f, indent=1)
    with open(os.path.join(REPO, args.out + '.md'), 'w') as f:
        f.write('\n'.join(txt) + '\n')
