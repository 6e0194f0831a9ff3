#!/usr/bin/env python3
"""bench.py — img/s of the YOLACT hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
    Yolact.forward(img[B,3,544,544])                         (HIP engine, hipGraph replay)
    + for every image: nms() + after_nms(480x640)            (HIP kernels; on the dense synthetic head
                                                              outputs of BASELINE.md §3, because a random-init
                                                              network produces degenerate detection counts)
Image size is 544 (the reference cannot run at 550: SURVEY.md §0.1).  Weights: seeded random init.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg res101_coco] [--batch 1]

`--inflight S` (default 4 at bs=1): S independent bs=1 requests are in flight on S HIP streams (S engines with their own activations,
split-K scratch, arrival counters and hipGraphs; GPU_MAX_HW_QUEUES=8 so that every stream has a hardware queue of its own).  A
bs=1 forward is a chain of ~190 dependent launches, each ~9 us of launch boundary + address set-up + epilogue around ~7 us of MFMA
work, so ONE chain keeps the matrix pipe ~35 % busy; the other requests' kernels run in those holes.  Measured on one MI355X
(mid-round 3; the end-of-round figures are 338 / 600), forward + nms + after_nms: 1 request 325 img/s, 2: 468, 3: 545, 4: 594, 5: 495, 6: 532, 8: 479 (the part runs four compute pipes).
Every step is still ONE image through forward + nms + after_nms with ONE host read of its detection count (taken when the slot is
reused, S steps later: the count is copied to pinned memory behind the request, so the host never waits on what it just
enqueued).  `--inflight 1` is the single-request latency mode of rounds 1-2; its numbers stay in the line under
`roofline.single_request`.

N > 1 (launched by torch.distributed.run, one rank per GPU): inference does not shard — images are
independent — so every rank runs an independent replica on its own batch ("replicas only", no data-path
collective); the timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

# requests in flight (--inflight) run on separate HIP streams; ROCm multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and two streams that share a queue do not overlap -> ask for 8 before the runtime starts
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--cfg', default='res101_coco')
    ap.add_argument('--batch', type=int, default=1, help='images per GPU per step')
    ap.add_argument('--img_size', type=int, default=544)
    ap.add_argument('--inflight', type=int, default=0, help='bs=1 requests in flight on separate streams (0 = 4 at --batch 1, else 1)')
    ap.add_argument('--no-post', action='store_true', help='time the network forward only')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the bs=8 side measurements')
    ap.add_argument('--lean', action='store_true', help='profiler runs (tools/profile_round.sh): no spread repeats, no latency leg')
    ap.add_argument('--mode', default='infer', choices=['infer', 'train'],
                    help="what `value` reports; the other mode's numbers still appear under extra")
    ap.add_argument('--train-batch', type=int, default=8, help='images per GPU per training step')
    ap.add_argument('--train-steps', type=int, default=6)
    ap.add_argument('--no-train', action='store_true', help='skip the DDP training measurement')
    ap.add_argument('--local_rank', type=int, default=None)
    ap.add_argument('--leg', default='', help=argparse.SUPPRESS)        # internal: one leg in a process of its own (prints a JSON dict)
    return ap.parse_args()


def build_net(cfg_name, img_size, device, seed=0):
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    cfg = build_cfg(cfg_name, 'val', img_size)
    torch.manual_seed(seed)
    net = Yolact(cfg).eval().to(device)
    return net, cfg


class Workload:
    """forward(batch) + per-image post-processing, everything resident on the device.  `inflight` > 1 (batch 1 only): that many
    requests overlap (yolact_minimal_amd.pipeline.RequestPipeline: one engine + one stream per slot); see the module docstring.
    `chained`: post-process the forward's OWN outputs (eval.py:45-52) instead of the dense synthetic head outputs."""

    def __init__(self, net, cfg, batch, img_size, device, with_post=True, seed=0, inflight=1, chained=False, timed=False):
        from yolact_minimal_amd.utils.synthetic import synth_head_outputs
        from yolact_minimal_amd.pipeline import RequestPipeline
        self.net, self.cfg, self.batch, self.device = net, cfg, batch, device
        g = torch.Generator().manual_seed(seed)
        self.img = torch.randn(batch, 3, img_size, img_size, generator=g).to(device)
        self.with_post = with_post
        n_anchors = len(net.anchors) // 4
        cls, box, coef, proto = synth_head_outputs(n_anchors, num_classes=cfg.num_classes, proto_hw=img_size // 4,
                                                   seed=1)
        self.head = None if chained else [t.to(device) for t in (cls, box, coef, proto)]
        # batch > 1: the batched launch set (one image per grid row, one host read per batch) on B copies of the dense case
        self.head_b = [t.expand(batch, *t.shape[1:]).contiguous() for t in self.head] if batch > 1 and not chained else None
        self.anchors = torch.tensor(net.anchors, dtype=torch.float32).reshape(-1, 4).to(device)
        self.inflight = inflight
        self.pipe = None
        if self.inflight > 1:
            self.pipe = RequestPipeline(net, cfg, img_size, img_size, device, depth=self.inflight, out_hw=(480, 640), with_post=with_post,
                                        batch=batch, return_outputs=False, timed=timed)
            self.pipe.warm_up(self.img, self.head if batch == 1 else self.head_b)   # (graph capture + 4 requests per slot: a warm allocator)
            self.engine = self.pipe.engines[0]
        else:
            self.engine = net._engine(self.img)

    def flush(self):
        if self.pipe is not None:
            self.pipe.drain()

    def step(self):
        from yolact_minimal_amd.utils.output_utils import nms, after_nms, nms_batch, after_nms_batch
        if self.pipe is not None:
            self.pipe.submit(self.img, self.head if self.batch == 1 else self.head_b)
            return
        self.engine.run(self.img)
        if self.with_post:
            if self.batch > 1:
                hb = self.head_b if self.head_b is not None else self.engine.outputs()
                after_nms_batch(nms_batch(*hb, self.anchors, self.cfg), 480, 640, self.cfg)
            else:
                cls, box, coef, proto = self.head if self.head is not None else self.engine.outputs()
                r = nms(cls, box, coef, proto, self.anchors, self.cfg)
                after_nms(r[0], r[1], r[2], r[3], r[4], 480, 640, self.cfg)


def timed(workload, steps, warmup, barrier):
    for _ in range(warmup):
        workload.step()
    workload.flush()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        workload.step()
    workload.flush()                 # (requests in flight: the host reads of the last ones)
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0


def conv_roofline(engine, img, iters=5):
    """Live HIP-event timing of every conv_igemm_f32 launch on the stream it is launched on (torch's current
    stream), eager (no graph) so each launch can be bracketed.  Returns (flops per forward, seconds per forward,
    launches per forward)."""
    from yolact_minimal_amd import hip
    # (the fused stem + max-pool launch counts as a conv launch: it carries the stem's FLOPs)
    timed = [(kind, arg) for kind, arg in engine.ops if kind in ('conv', 'stem_pool')]
    convs = [arg if kind == 'conv' else arg[0] for kind, arg in timed]
    ws = engine.workspace
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in convs]
    total = 0.0
    per_layer = [0.0] * len(convs)
    for it in range(iters + 1):
        if engine.fused_stem is None:
            hip.nchw_to_nhwc4(img, engine.x_in)
        engine._img = img
        ci = 0
        for kind, arg in engine.ops:
            if kind in ('conv', 'stem_pool'):
                evs[ci][0].record()
                engine._launch_one(kind, arg, ws)
                evs[ci][1].record()
                ci += 1
            else:
                engine._launch_one(kind, arg, ws)
        torch.cuda.synchronize()
        if it == 0:
            continue   # warm-up
        for i, (a, b) in enumerate(evs):
            ms = a.elapsed_time(b)
            per_layer[i] += ms
            total += ms
    secs = total / iters / 1e3
    flops = sum(c.flops for c in convs)
    layers = [dict(name=c.name, ms=per_layer[i] / iters, gflop=c.flops / 1e9) for i, c in enumerate(convs)]
    return flops, secs, len(convs), layers


def post_bench(net, cfg, device, img_size, iters=30):
    """nms and after_nms on the dense synthetic head outputs (17.8 k candidates -> 100 detections), timed with HIP events on the
    launch stream.  after_nms is HBM-bound on its output: n * img_h * img_w * 4 bytes written once (123 MB at 100 x 480 x 640)."""
    from yolact_minimal_amd.utils.output_utils import nms, after_nms, nms_batch, after_nms_batch
    from yolact_minimal_amd.utils.synthetic import synth_head_outputs
    n_anchors = len(net.anchors) // 4
    head = [t.to(device) for t in synth_head_outputs(n_anchors, num_classes=cfg.num_classes, proto_hw=img_size // 4, seed=1)]
    anchors = torch.tensor(net.anchors, dtype=torch.float32).reshape(-1, 4).to(device)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    out = {}
    for b in (1, 8):
        hb = [t.expand(b, *t.shape[1:]).contiguous() for t in head]
        res = {}
        # 'eager': the stream is idle when the first event is recorded, so nms_us also holds the host's way from the event to the
        # first launch (Python + ctypes); 'queued': a ~1 ms spin kernel in front, the launches wait in the stream behind it -- what the
        # DEVICE spends on nms (the request graphs replay it like that; tools/nms_span.py reads the same span from a kernel trace)
        for mode in ('eager', 'queued'):
            t_nms = t_after = 0.0
            for it in range(iters + 3):
                if mode == 'queued':
                    torch.cuda._sleep(2_000_000)
                e[0].record()
                dets = nms_batch(*hb, anchors, cfg)
                e[1].record()
                r = after_nms_batch(dets, 480, 640, cfg, sync=False)
                e[2].record()
                torch.cuda.synchronize()
                if it >= 3:
                    t_nms += e[0].elapsed_time(e[1])
                    t_after += e[1].elapsed_time(e[2])
            res[mode] = (t_nms / iters * 1e-3, t_after / iters * 1e-3)
        n_det = int(r[4].sum())
        nbytes = n_det * 480 * 640 * 4
        (t_nms, t_after), (q_nms, q_after) = res['eager'], res['queued']
        out[f'bs{b}'] = dict(detections=n_det, nms_us=round(t_nms * 1e6, 1), after_nms_us=round(t_after * 1e6, 1),
                             nms_device_us=round(q_nms * 1e6, 1), after_nms_device_us=round(q_after * 1e6, 1),
                             after_nms_gbs=round(nbytes / t_after / 1e9, 1), after_nms_frac_hbm_peak=round(nbytes / t_after / 1e9 / HBM_PEAK_GBS, 4))
    return out


def detecting_net(cfg_name, img_size, device):
    """A random-init network gives degenerate detections, so the shared conf layer is reshaped like oracle/make_golden_chained.py
    does (weight x 10, bias = minus the spatial mean of every (anchor, class) logit + N(0, 1) + a background offset found by
    bisection for ~400 candidates over the score threshold); the golden test of that recipe is
    tests/test_gpu_pipeline.py::test_chained_forward_nms_after_nms_matches_the_reference.  Returns (net, cfg, img)."""
    net, cfg = build_net(cfg_name, img_size, device, seed=71)
    conv = net.prediction_layers.conf_layer
    with torch.no_grad():
        conv.weight.mul_(10.0)
        conv.bias.zero_()
    net.mark_weights_changed()
    img = torch.randn(1, 3, img_size, img_size, generator=torch.Generator().manual_seed(371)).to(device)
    eng = net._engine(img)
    eng.run(img)
    torch.cuda.synchronize()
    la = eng.class_logits[0].reshape(-1, 3, cfg.num_classes).double()
    nb = torch.randn(3, cfg.num_classes, generator=torch.Generator().manual_seed(571)).to(device).double() - la.mean(dim=0)
    lo, hi = 0.0, 40.0
    for _ in range(30):
        mid = 0.5 * (lo + hi)
        bb = nb.clone()
        bb[:, 0] += mid
        n_c = int((torch.softmax((la + bb).reshape(-1, cfg.num_classes), -1)[:, 1:].max(dim=1)[0] > 0.05).sum())
        lo, hi = (mid, hi) if n_c > 400 else (lo, mid)
    nb[:, 0] += hi
    with torch.no_grad():
        conv.bias.copy_(nb.reshape(-1).float())
    net.mark_weights_changed()
    net._engines.clear()
    return net, cfg, img


def chained_bench(cfg_name, img_size, device, inflight, steps=100):
    """eval.py:45-52 as ONE chain: `nms` / `after_nms` consume the forward's OWN outputs (network: `detecting_net`)."""
    net, cfg, img = detecting_net(cfg_name, img_size, device)
    out = {}
    for s_ in sorted({1, inflight}):
        w = Workload(net, cfg, 1, img_size, device, with_post=True, inflight=s_, chained=True)
        t = timed(w, steps, 10, lambda: None) / steps
        out[f'img_s_inflight{s_}'] = round(1.0 / t, 1)
        if w.pipe is not None:
            out['detections_per_image'] = round(w.pipe.detections / max(1, w.pipe.submitted), 1)
        del w
    out['workload'] = 'forward + nms + after_nms(480x640) on the forward\'s own outputs (~400 candidates over the score threshold)'
    net._engines.clear()
    return out


def _dropin_loops():
    """dropin/reference_loops.py (the statements of the reference's train.py / eval.py), imported the way dropin/run.py binds them."""
    for q in (os.path.join(REPO, 'dropin'),):
        if q not in sys.path:
            sys.path.insert(0, q)
    import reference_loops
    return reference_loops


def eval_loop_bench(cfg_name, img_size, device, images=40, h=480, w=640):
    """The reference-shaped evaluation loop (eval.py:36-69 through `dropin/`: forward -> nms -> after_nms -> metric, ONE image at a
    time, every stage fenced by a device synchronize as `utils.timer.counter` does) on `detecting_net`, 15 synthetic ground-truth
    instances per image.  Three metric branches: `prep_metrics` (eval.py:69, tensors stay on the device: the default), `--coco_api`
    as the reference writes it (eval.py:60-67: boxes + the DENSE fp32 masks cross PCIe, then one `add_mask` per detection), and the
    same records with device-side RLE (`coco_api='device'`: no dense D2H).  `value` of the headline line is NOT this loop: it
    keeps 4 requests in flight and no host read of masks."""
    from yolact_minimal_amd.utils.synthetic import synth_eval_case
    L = _dropin_loops()
    net, cfg, img = detecting_net(cfg_name, img_size, device)
    _, _, _, _, gt, gt_masks, _, _ = synth_eval_case(1, 40, 15, h, w, 10)
    gt, gt_masks = gt.to(device), gt_masks.to(device)
    out = dict(workload=f'{cfg_name} {img_size}x{img_size}, one image at a time, after_nms at {h}x{w}, 15 gt instances per image; inputs '
                        f'resident in HBM (img.cuda() is a no-op), statements of eval.py:36-69')

    def loader(n):
        return [(img, gt.clone(), gt_masks, h, w) for _ in range(n)]
    for key, kw, n in (('prep_metrics', dict(coco_api=False), images), ('prep_metrics_no_stage_fences', dict(coco_api=False, sync_stages=False), images),
                       ('coco_api_device_rle', dict(coco_api='device'), images), ('coco_api_dense_masks_over_pcie', dict(coco_api=True), max(4, images // 5))):
        # (the dense branch last: a process uses ONE metric branch, and after the 123 MB pageable D2H copies of the dense one the small
        #  host reads of the next leg were 3x slower -- 1.6 instead of 0.47 ms of metric stage for the device-RLE leg in one process)
        if os.environ.get('YM_EVAL_LEGS') and key not in os.environ['YM_EVAL_LEGS'].split(','):
            continue
        L.eval_loop(net, cfg, loader(3), **kw)                        # warm-up (plans, graph capture, allocator)
        _, mj, seen, secs = L.eval_loop(net, cfg, loader(n), **kw)
        row = dict(img_s=round(n / secs, 1), ms_per_img=round(secs / n * 1e3, 3), images=n, images_with_detections=seen)
        if kw.get('sync_stages', True):
            t = L.timer.get_times(['forward', 'nms', 'after_nms', 'metric'])
            row['stage_ms'] = dict(forward=round(t[0] * 1e3, 3), nms=round(t[1] * 1e3, 3), after_nms=round(t[2] * 1e3, 3), metric=round(t[3] * 1e3, 3))
        if mj is not None:
            row['records_per_image'] = round(len(mj.mask_data) / max(1, n), 1)
        out[key] = row
    # what bounds the dense branch: the D2H of the image's masks (pageable destination, as `.cpu()` allocates it)
    n_det = out.get('coco_api_dense_masks_over_pcie', {}).get('records_per_image') or 100
    m = torch.zeros(int(round(n_det)), h, w, device=device)
    m.cpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.cpu()
    t_d2h = (time.perf_counter() - t0) / 5
    out['dense_mask_d2h'] = dict(mb_per_image=round(m.numel() * 4 / 1e6, 1), ms=round(t_d2h * 1e3, 3), gb_s=round(m.numel() * 4 / t_d2h / 1e9, 2),
                                 note='`masks_p.cpu()` of eval.py:62 alone (pageable host memory): the floor of the dense --coco_api branch')
    out['bounds'] = ('prep_metrics: host-side stage fences + python around ~2.5 ms of device work per image; coco_api dense: PCIe D2H of '
                     'n x H x W fp32 masks, then one RLE call per mask; coco_api device RLE: one ym_rle_encode launch + a few KB D2H per image')
    net._engines.clear()
    return out


def train_reference_loop_bench(cfg_name, img_size, batch, steps, warmup, local_rank, device):
    """The reference's OWN training step on the HIP path (train.py:60-63,76,102-130 through `dropin/`): `optim.SGD(net.parameters())`
    (AdamW for swin_tiny_coco), `DDP(net.cuda(), [local_rank], output_device=local_rank, broadcast_buffers=True)` on a one-rank RCCL
    group, `net(images, targets, masks)`, `dist.all_reduce(all_loss)`, `zero_grad()`, `loss_total.backward()`, `optimizer.step()` —
    not the build's `Trainer` (`extra.train`).  Same synthetic batch as `train_bench`; bit-identical results to `Trainer.step` are
    tests/test_gpu_reference_loop.py's business.  What it lacks against `Trainer`: gradients land in fresh tensors that DDP copies
    into its buckets (no flat buffer), weight gradients stay on the main stream (their side stream needs the flat slots), one
    foreach-SGD kernel group instead of one launch."""
    import torch.distributed as dist
    from yolact_minimal_amd.utils.synthetic import synth_targets
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    L = _dropin_loops()
    own_group = not dist.is_initialized()
    if own_group:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend='nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        cfg = build_cfg(cfg_name, 'train', img_size, train_bs=batch, bs_per_gpu=batch)
        torch.manual_seed(0)
        net = Yolact(cfg)
        net.train()
        optimizer = L.make_optimizer(net, cfg)
        net = L.wrap_ddp(net, local_rank)
        g = torch.Generator().manual_seed(100)
        img = torch.randn(batch, 3, img_size, img_size, generator=g)
        boxes, masks = synth_targets(batch, img_size, seed=0)
        res, stages = {}, {}
        last = []
        for key, to_dev in (('inputs_resident', True), ('inputs_from_host', False)):
            im = img.to(device) if to_dev else img
            bx = [b.to(device) if to_dev else b for b in boxes]
            mk = [m.to(device) if to_dev else m for m in masks]

            def loader(n):
                return ((im, [b.clone() for b in bx], mk) for _ in range(n))
            marks = []

            def keep(step, losses, lr):          # (no host read per step)
                last[:] = [losses]
                marks.append(time.perf_counter())
            # ONE call of the loop, like train.py: its timer starts at the second iteration, and from then on every phase of a step
            # is fenced by `timer.counter`'s device synchronizes (utils/timer.py:63-76) -- the published loop never overlaps phases
            L.train_loop(net, optimizer, cfg, loader(warmup + steps), max_steps=warmup + steps, on_step=keep)
            torch.cuda.synchronize()
            res[key] = (time.perf_counter() - marks[warmup - 1]) / steps
            stages[key] = [round(v * 1e3, 2) for v in L.timer.get_times(['for+loss', 'backward', 'update'])]
        # the same loop with the timer left stopped (no fences): what the loop could do if it did not time its phases
        marks = []
        L.train_loop(net, optimizer, cfg, ((img.to(device), [b.to(device) for b in boxes], [m.to(device) for m in masks]) for _ in range(warmup + steps)),
                     max_steps=warmup + steps, on_step=lambda *a: marks.append(time.perf_counter()), fences=False)
        torch.cuda.synchronize()
        t_nofence = (time.perf_counter() - marks[warmup - 1]) / steps
        t = res['inputs_resident']
        flops_img = 3.0 * {'res101': 157.2e9, 'res50_': 113.4e9, 'swin_t': 119.2e9}.get(cfg_name[:6], 157.2e9)
        losses = last[0]
        return dict(img_s=round(batch / t, 2), ms_per_step=round(t * 1e3, 2), steps=steps, warmup=warmup, batch_per_gpu=batch,
                    optimizer=type(optimizer).__name__, wrapper=type(net).__name__, world_size=dist.get_world_size(), backend=dist.get_backend(),
                    frac_f32_mfma_peak=round(batch / t * flops_img / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                    stage_ms=dict(zip(('for+loss', 'backward', 'update'), stages['inputs_resident'])),
                    without_timer_fences=dict(img_s=round(batch / t_nofence, 2), ms_per_step=round(t_nofence * 1e3, 2)),
                    module_train_state=getattr(net.module, '_train_state', None) is not None,
                    inputs_from_host=dict(img_s=round(batch / res['inputs_from_host'], 2), ms_per_step=round(res['inputs_from_host'] * 1e3, 2),
                                          note='CPU tensors handed to train.py:112-114 (`images.cuda()`: pageable H2D of images + masks inside the step)'),
                    last_losses=[round(float(l.detach()), 4) for l in losses], finite=all(bool(torch.isfinite(l)) for l in losses),
                    note='train.py\'s own statements (torch DDP + torch.optim, phases fenced by its timer) on the HIP autograd path; the module brings '
                         'its own flat gradient slots / side stream / reducer (train_state.py; round 5, without them: 72.3 ms per step). '
                         '`extra.train` is the build\'s Trainer')
    finally:
        if own_group:
            dist.destroy_process_group()


def other_sizes_bench(cfg_name, device, sizes=(320, 544, 736)):
    """Any multiple of 32 is a valid `--img_size` (config.py:75, detect.py:24, eval.py:18); the tuned table is keyed on exact layer
    shapes.  Forward-only, bs=1, one request at a time (graph replay), per size under three plan sources: `<size>px` = the shipped
    table (rows for 256 ... 800 px; a shape without a row takes the nearest tuned shape's row, plan_transfer.py), `<size>px_transfer`
    = transfers ONLY (`YM_TUNED_NEAREST=only`: every launch on a row re-derived from another shape -- what an unmeasured size
    gets), `<size>px_fallback` = the planner heuristic without any row (what every other size ran on before round 6)."""
    from yolact_minimal_amd import engine as E
    out = {}

    def run(size):
        net, cfg = build_net(cfg_name, size, device)
        w = Workload(net, cfg, 1, size, device, with_post=False)
        t = min(timed(w, 60, 5, lambda: None), timed(w, 60, 0, lambda: None)) / 60
        fl = w.engine.total_flops
        src = {}
        for c in w.engine.convs:
            k = c.plan_source.split(':')[0]
            src[k] = src.get(k, 0) + 1
        return dict(forward_ms=round(t * 1e3, 3), img_s=round(1.0 / t, 1), gflop_per_img=round(fl / 1e9, 1),
                    frac_f32_mfma_peak=round(fl / t / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), conv_launches=len(w.engine.convs), plan_sources=src)
    saved, saved_env = E._tuned, os.environ.get('YM_TUNED_NEAREST')
    try:
        for size in sizes:
            out[f'{size}px'] = run(size)
            os.environ['YM_TUNED_NEAREST'] = 'only'
            out[f'{size}px_transfer'] = run(size)
            os.environ.pop('YM_TUNED_NEAREST')
            E._tuned = {}
            out[f'{size}px_fallback'] = run(size)
            E._tuned = saved
    finally:
        E._tuned = saved
        if saved_env is None:
            os.environ.pop('YM_TUNED_NEAREST', None)
        else:
            os.environ['YM_TUNED_NEAREST'] = saved_env
    out['544px_tuned'] = out['544px']           # (the key of earlier rounds' lines)
    out['note'] = ('same network and kernels per size; only the per-shape tile / split / kernel-family choice differs between the three '
                   'plan sources')
    return out


def eval_metrics_bench(device, n=100, g=15, h=480, w=640, iters=20, cpu=True):
    """Next-row f2: `prep_metrics` (mask IoU 100 x 15 masks at 480x640 + box IoU + matching over 10 thresholds) per image.
    HBM roofline of the mask-IoU kernel: every mask is read once = (n + g) * h * w * 4 bytes."""
    from yolact_minimal_amd.utils.synthetic import synth_eval_case
    from yolact_minimal_amd.utils import common_utils as C
    from yolact_minimal_amd.utils.box_utils import mask_iou
    thres = [x / 100 for x in range(50, 100, 5)]
    ids, scores, boxes, masks, gt, gt_masks, h, w = synth_eval_case(1, n, g, h, w, 10)
    d = [boxes.to(device), masks.to(device), gt.to(device), gt_masks.to(device)]
    a, b = d[1].reshape(n, -1), d[3].reshape(g, -1)
    for _ in range(3):
        mask_iou(a, b, to_cpu=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        mask_iou(a, b, to_cpu=False)
    e1.record()
    torch.cuda.synchronize()
    t_iou = e0.elapsed_time(e1) / iters * 1e-3
    # the same two launches through the C ABI with the scratch / output allocated once (what `mask_iou` costs on the device: the
    # python wrapper's two allocations per call are host time that back-to-back eager calls expose)
    import ctypes
    from yolact_minimal_amd import hip as H
    nb = H.lib().ym_mask_iou_workspace_bytes(n, g, a.shape[1])
    ws, iou = torch.empty(nb, device=device, dtype=torch.uint8), torch.empty(n, g, device=device)
    t_dev = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            H.lib().ym_mask_iou(H.ptr(a), n, H.ptr(b), g, a.shape[1], H.ptr(iou), ctypes.c_void_p(ws.data_ptr()), nb, H.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        t_dev = min(t_dev, e0.elapsed_time(e1) / iters * 1e-3)
    t0 = time.perf_counter()
    for _ in range(iters):
        ap = {k: [[C.APDataObject() for _ in range(10)] for _ in thres] for k in ('box', 'mask')}
        C.prep_metrics(ap, ids, scores, d[0], d[1], d[2].clone(), d[3], h, w, thres)
    torch.cuda.synchronize()
    t_prep = (time.perf_counter() - t0) / iters
    nbytes = (n + g) * h * w * 4
    out = dict(workload=f'{n} predicted x {g} gt masks at {h}x{w}, 10 IoU thresholds', mask_iou_us=round(t_iou * 1e6, 1),
               mask_iou_gbs=round(nbytes / t_iou / 1e9, 1), frac_hbm_peak=round(nbytes / t_iou / 8e12, 4),
               mask_iou_device_us=round(t_dev * 1e6, 1), mask_iou_device_gbs=round(nbytes / t_dev / 1e9, 1),
               mask_iou_device_frac_hbm_peak=round(nbytes / t_dev / 8e12, 4), prep_metrics_ms=round(t_prep * 1e3, 3))
    # next-row f3: COCO RLE of the same 100 masks (two passes over each mask = 2 * n*h*w*4 bytes; strings cross PCIe)
    for _ in range(2):
        rles = C.rle_encode(d[1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        rles = C.rle_encode(d[1])
    t_rle = (time.perf_counter() - t0) / iters
    out.update(rle_encode_ms=round(t_rle * 1e3, 3), rle_gbs=round(2 * n * h * w * 4 / t_rle / 1e9, 1),
               rle_bytes_per_image=sum(len(r['counts']) for r in rles), dense_mask_bytes_per_image=n * h * w * 4)
    if cpu:
        from oracle import metrics_ref as M, rle_ref as R       # cpu_baseline leg only
        t0 = time.perf_counter()
        for _ in range(3):
            ref = M.new_ap_data(10, len(thres))
            M.prep_metrics(ref, ids, scores, boxes, masks, gt, gt_masks, h, w, thres)
        out['cpu_oracle_prep_metrics_ms'] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
        mk = masks.numpy()
        t0 = time.perf_counter()
        ref_rle = [R.encode(mk[i]) for i in range(n)]
        out['cpu_oracle_rle_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
        out['rle_equal_to_oracle'] = ref_rle == rles
    return out


def train_aug_bench(device, iters=30, cpu=True):
    """Next-row f4: `train_aug` of one COCO-like sample (480x640 uint8 image, 6 instance masks) to the 544 train size.
    HBM view: the image + masks are read about once and [3 + k, 544, 544] floats are written."""
    import random
    from yolact_minimal_amd.utils.synthetic import synth_sample
    from yolact_minimal_amd.utils.augmentations import train_aug
    img, masks, boxes, labels = synth_sample(12, 480, 640, 6)
    g_img, g_masks = torch.from_numpy(img).to(device), torch.from_numpy(masks).to(device)
    random.seed(3)
    for _ in range(3):
        train_aug(g_img, g_masks, boxes, labels, 544)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    for _ in range(iters):
        done += train_aug(g_img, g_masks, boxes, labels, 544)[0] is not None
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    out = dict(workload='480x640 uint8 image + 6 masks -> 544x544 (photometric, mirror, crop, pad, resize, pad/crop, normalise)',
               ms_per_sample=round(t * 1e3, 3), samples_per_s=round(1.0 / t, 1), accepted=done)
    if cpu:
        from oracle import augment_ref as A
        from yolact_minimal_amd.utils.augmentations import sample_train_aug
        random.seed(3)
        t0 = time.perf_counter()
        n = 0
        for _ in range(5):
            plan = sample_train_aug(480, 640, boxes, labels, 544)
            if plan is not None:
                A.apply_plan(img, masks, plan)
                n += 1
        out['cpu_oracle_ms_per_sample'] = round((time.perf_counter() - t0) / max(n, 1) * 1e3, 2)
    return out


def ann_to_mask_bench(device, iters=30, cpu=True):
    """Next-row f4 (reader): the polygon annotations of one COCO-like image (480x640, 7 instances of 1-3 polygons) -> dense uint8
    masks (`COCO.annToMask` per annotation in the reference).  HBM view: n*H*W bytes written once; the bitmaps live in LDS."""
    from yolact_minimal_amd.utils.synthetic import synth_polygons
    from yolact_minimal_amd.utils.coco import anns_to_masks
    h, w = 480, 640
    segs = synth_polygons(21, h, w, n=7)
    for _ in range(3):
        anns_to_masks(segs, h, w, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        anns_to_masks(segs, h, w, device)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    out = dict(workload='7 annotations (1-3 polygons each, 3-40 vertices) -> [7, 480, 640] uint8 masks, incl. the H2D of the vertices',
               ms_per_image=round(t * 1e3, 3), images_per_s=round(1.0 / t, 1))
    if cpu:
        from oracle.coco_ref import segm_to_mask as _segm_to_mask      # cpu_baseline leg only
        t0 = time.perf_counter()
        for s in segs:
            _segm_to_mask(s, h, w)
        out['cpu_oracle_ms_per_image'] = round((time.perf_counter() - t0) * 1e3, 2)
    return out


def train_to_map_bench():
    """Train -> evaluate -> mAP on the synthetic shapes dataset (tools/overfit_demo.py: 600 `Trainer` steps on a random-init
    res50_custom at 128 px, then nms -> after_nms -> prep_metrics -> calc_map on the training pictures), with the numbers of the
    REAL reference's own run of the same recipe beside it (frozen by oracle/overfit_reference.py as a golden fixture: data, not
    the oracle).  The offline stand-in for the north star's mAP line."""
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    from overfit_demo import run
    ref = json.load(open(os.path.join(REPO, 'tests', 'golden', 'overfit_reference_128.json')))
    got = run(steps=ref['steps'], n_images=ref['images'], size=ref['size'], batch=ref['batch'], cfg_name=ref['cfg'], seed=ref['seed'],
              log=lambda *_: None, log_every=100)
    return dict(workload=f"{ref['cfg']} {ref['size']} px bs={ref['batch']}, {ref['steps']} steps on {ref['images']} synthetic pictures, "
                         'scored on the same pictures (eval.py:36-110 statements)',
                train_s=got['train_s'], steps_per_s=round(ref['steps'] / got['train_s'], 1),
                box_map=got['box_map'][0], mask_map=got['mask_map'][0], box_map50=got['box_map'][1], mask_map50=got['mask_map'][1],
                box_map_traditional_nms=got['box_map_traditional_nms'][0], mask_map_traditional_nms=got['mask_map_traditional_nms'][0],
                serving_path_identical_pictures=got['serving_path_identical_pictures'], detections=got['detections'],
                first_step_losses=got['losses'][0][1], last_step_losses=got['losses'][-1][1],
                reference_cpu=dict(box_map=ref['box_map'][0], mask_map=ref['mask_map'][0], box_map50=ref['box_map'][1],
                                   mask_map50=ref['mask_map'][1], first_step_losses=ref['losses'][0][1],
                                   last_step_losses=ref['losses'][-1][1], train_s=ref['cpu_s'],
                                   source='tests/golden/overfit_reference_128.json (the real reference, imported, CPU)'))


def train_bench(cfg_name, img_size, batch, steps, warmup, world, local_rank, device, barrier):
    """DDP training: one step = forward + loss + backward (+ RCCL gradient all-reduce overlapped by DDP hooks) +
    SGD step on `batch` synthetic images per GPU (targets: 4 boxes + rectangular masks per image)."""
    from yolact_minimal_amd.utils.synthetic import synth_targets
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    from yolact_minimal_amd.trainer import Trainer, reduce_max
    cfg = build_cfg(cfg_name, 'train', img_size, train_bs=batch * world, bs_per_gpu=batch)
    torch.manual_seed(0)
    net = Yolact(cfg)
    tr = Trainer(net, cfg, device, world, local_rank)
    rank = int(os.environ.get('RANK', '0'))
    g = torch.Generator().manual_seed(100 + rank)
    img = torch.randn(batch, 3, img_size, img_size, generator=g).to(device)
    boxes, masks = synth_targets(batch, img_size, seed=1000 * rank)
    boxes, masks = [b.to(device) for b in boxes], [m.to(device) for m in masks]
    losses = None
    for _ in range(warmup):
        losses = tr.step(img, boxes, masks)
    tr.enable_timing()                     # three HIP event pairs per step (buffer broadcast, backward end -> last bucket reduced)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = tr.step(img, boxes, masks)
    torch.cuda.synchronize()
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0, device)
    timing = tr.timing_summary()
    flops_img = 3.0 * {'res101': 157.2e9, 'res50_': 113.4e9, 'swin_t': 119.2e9}.get(cfg_name[:6], 157.2e9)   # step ~ 3x forward
    opt = 'AdamW' if cfg_name.startswith('swin') else 'SGD'
    img_s = batch * world * steps / elapsed
    out = dict(img_s=round(img_s, 2), ms_per_step=round(elapsed / steps * 1e3, 2), steps=steps, warmup=warmup,
               batch_per_gpu=batch, global_batch=batch * world, parallelism=f'ddp{world} (RCCL all-reduce, 25 MB buckets)', optimizer=opt,
               tflops_per_gpu=round(img_s / world * flops_img / 1e12, 2),
               frac_f32_mfma_peak=round(img_s / world * flops_img / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
               last_losses=[round(float(l.detach()), 4) for l in losses], finite=all(bool(torch.isfinite(l)) for l in losses))
    if tr.ddp:
        out['ddp'] = ddp_block(tr, timing, img_s, world, cfg_name, batch)
    return out


def ddp_block(tr, timing, img_s, world, cfg_name, batch):
    """The north star's DDP figures (reference train.py:70-77,122; README.md:54-57), readable from the record alone: training
    img/s of the whole job and per GPU, the 1-GPU reference it scales against (the committed N = 1 line of this round), what the
    collectives cost where backward could not hide them, and what the process group actually saw."""
    import torch.distributed as dist
    ref, ref_src = None, None
    for r in (6, 5, 4, 3):
        q = os.path.join(REPO, 'profiles', f'r0{r}_bench_line.json')
        if os.path.exists(q):
            try:
                tb = json.load(open(q))['extra']['train' if batch != 16 else 'train_bs16']
                if tb.get('batch_per_gpu') == batch and cfg_name == 'res101_coco':
                    ref, ref_src = float(tb['img_s']), os.path.relpath(q, REPO)
                    break
            except (KeyError, ValueError, TypeError):
                pass
    try:
        nccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:                                   # (gloo control-flow tests)
        nccl = None
    red = tr.reducer
    log = getattr(red, 'last_launch_log', []) if red is not None else []
    return dict(train_img_s=round(img_s, 2), per_gpu=round(img_s / world, 2), vs_1gpu_reference_img_s=ref,
                scaling_vs_1gpu=round(img_s / ref, 3) if ref else None, reference_source=ref_src,
                world_size_seen=dist.get_world_size() if dist.is_initialized() else 1,
                backend=dist.get_backend() if dist.is_initialized() else None, rccl_version=nccl,
                buckets=len(red.buckets) if red is not None else None,
                bucket_mb=[round((e - a) * 4 / 2 ** 20, 1) for a, e, _ in red.buckets] if red is not None else None,
                buckets_launched_during_backward=sum(1 for _, _, in_finish in log if not in_finish),
                allreduce_exposed_ms=timing.get('allreduce_exposed_ms'), buffer_broadcast_ms=timing.get('buffer_broadcast_ms'),
                backward_ms=timing.get('backward_ms'),
                note='allreduce_exposed_ms: end of backward on the device (both streams joined) -> last bucket reduced, from HIP events '
                     'on the step\'s stream; buffer_broadcast_ms: the one-message BN running-stat broadcast of the step')


def cpu_baseline(cfg_name, img_size, threads=None, budget_s=12.0, max_img=10):
    """The CPU oracle (plain PyTorch-CPU restatement of the reference) timed on this host's cores, bounded sample.
    `threads`: torch intra-op threads for the sample (None = torch's default = all host cores)."""
    from oracle import yolact_ref as R
    from yolact_minimal_amd.config import build_cfg
    from yolact_minimal_amd.modules.yolact import Yolact
    prev = torch.get_num_threads()
    if threads:
        torch.set_num_threads(threads)
    try:
        cfg = build_cfg(cfg_name, 'val', img_size)
        torch.manual_seed(0)
        sd = Yolact(cfg).eval().state_dict()
        img = torch.randn(1, 3, img_size, img_size, generator=torch.Generator().manual_seed(0))
        n_anchors = sum(((img_size + s - 1) // s) ** 2 * 3 for s in (8, 16, 32, 64, 128))
        cls, box, coef, proto = R.synth_head_outputs(n_anchors, proto_hw=img_size // 4, seed=1)
        anchors = R.anchors_for(img_size, cfg.scales)
        n_img, t_fwd, t_nms, t_after, best = 0, 0.0, 0.0, 0.0, 1e30
        t_start = time.perf_counter()
        with torch.no_grad():
            R.forward_eval_any(img, sd)                   # warm-up
            while n_img < 3 or (time.perf_counter() - t_start < budget_s and n_img < max_img):
                t0 = time.perf_counter(); R.forward_eval_any(img, sd); t1 = time.perf_counter()
                r = R.nms(cls, box, coef, proto, anchors); t2 = time.perf_counter()
                R.after_nms(r[0], r[1], r[2], r[3], r[4], 480, 640); t3 = time.perf_counter()
                t_fwd += t1 - t0; t_nms += t2 - t1; t_after += t3 - t2
                best = min(best, t3 - t0)
                n_img += 1
        nthr = torch.get_num_threads()
    finally:
        torch.set_num_threads(prev)
    # the host of a GPU box is shared and noisy (3x swings between runs were observed): the fastest image of the sample is the
    # baseline, the means are kept in the sample text
    return dict(value=round(1.0 / best, 3), unit='img/s', cores=nthr, kind='port',
                sample=f'best of {n_img} images bs=1 {cfg_name}@{img_size} ({best * 1e3:.0f} ms); means: oracle forward {t_fwd / n_img * 1e3:.0f} ms + nms '
                       f'{t_nms / n_img * 1e3:.0f} ms + after_nms(480x640) {t_after / n_img * 1e3:.0f} ms on '
                       f'{os.cpu_count()} host cpus ({nthr} torch threads)')


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec this command line under torch.distributed.run, one rank per GPU
    (the environment tools/launch_8gpu.sh sets).  The ranks' stdout is inherited, so rank 0's ONE JSON line is this process's."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # the host driver only supports dmabuf IPC
    env.setdefault('NCCL_MIN_NCHANNELS', '16')             # ring all-reduce over point-to-point xGMI is per-link bound
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    port = env.get('MASTER_PORT')
    if not port:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.leg == 'eval_loop':
        torch.cuda.set_device(0)
        if os.environ.get('YM_GC_OFF') == '1':        # (experiment: the collector off altogether)
            import gc
            gc.disable()
        else:                                          # the launcher's process policy (dropin/run.py::host_gc_policy; YM_DROPIN_GC=0: defaults)
            sys.path.insert(0, os.path.join(REPO, 'dropin'))
            import run as dropin_run
            dropin_run.host_gc_policy()
        fd = os.dup(1)
        os.dup2(2, 1)
        out = eval_loop_bench(args.cfg, args.img_size, torch.device('cuda', 0))
        os.write(fd, (json.dumps(out) + '\n').encode())
        return
    if args.leg == 'train_reference_loop':
        torch.cuda.set_device(0)
        sys.path.insert(0, os.path.join(REPO, 'dropin'))
        import run as dropin_run
        dropin_run.host_gc_policy()                    # (what `dropin/run.py train.py` does before the script; YM_DROPIN_GC=0: defaults)
        fd = os.dup(1)
        os.dup2(2, 1)
        out = train_reference_loop_bench(args.cfg, args.img_size, args.train_batch, args.train_steps, 2, 0, torch.device('cuda', 0))
        os.write(fd, (json.dumps(out) + '\n').encode())
        return
    if args.leg == 'train_to_map':
        torch.cuda.set_device(0)
        fd = os.dup(1)
        os.dup2(2, 1)
        out = train_to_map_bench()
        os.write(fd, (json.dumps(out) + '\n').encode())
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch one rank per GPU, or run bare and let '
                         f'bench.py spawn them)')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', args.local_rank or 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    local_rank %= torch.cuda.device_count()          # (tests run two gloo ranks on a 1-GPU box; one rank per GPU otherwise)
    # stdout carries ONE line, the JSON record: everything else a library prints there (RCCL's version banner, written through C
    # stdio and flushed at exit, i.e. AFTER the record) goes to stderr -- file descriptor 1 is re-pointed for the whole run and the
    # record is written to the saved descriptor at the end
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    if world > 1 or os.environ.get('YM_FORCE_DIST', '0') == '1':
        import torch.distributed as dist
        # nccl == RCCL on ROCm; YM_DIST_BACKEND=gloo only for the 2-ranks-on-one-GPU control-flow test
        dist.init_process_group(backend=os.environ.get('YM_DIST_BACKEND', 'nccl'), init_method='env://')

    def barrier():
        if dist is not None:
            dist.barrier()

    net, cfg = build_net(args.cfg, args.img_size, device)
    inflight = args.inflight if args.inflight > 0 else (4 if args.batch == 1 else 2)
    wl = Workload(net, cfg, args.batch, args.img_size, device, with_post=not args.no_post, inflight=inflight)
    elapsed_local = timed(wl, args.steps, args.warmup, barrier)
    elapsed = elapsed_local
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    imgs = args.batch * args.steps * world
    value = imgs / elapsed
    region_flops = wl.engine.total_flops * args.steps          # conv flops this rank executed inside the region `value` times
    # spread: the driver's sample is K steps (20 steps = 33 ms at bs=1) -> the same region five more times, and five regions of
    # >= 200 steps (this rank's own clock; every rank takes part so that the barriers match)
    nrep = 1 if args.lean else 5
    rep_k = [timed(wl, args.steps, 0, barrier) / args.steps * 1e3 for _ in range(nrep)]
    long_steps = args.steps if args.lean else max(args.steps, 200 // max(1, args.batch))
    rep_long = [timed(wl, long_steps, 0, barrier) / long_steps * 1e3 for _ in range(nrep)]
    spread = dict(unit='ms_per_step', region_steps=args.steps, first_region=round(elapsed_local / args.steps * 1e3, 4),
                  repeats=[round(x, 4) for x in rep_k], min=round(min(rep_k), 4), median=round(sorted(rep_k)[len(rep_k) // 2], 4), max=round(max(rep_k), 4),
                  long_region_steps=long_steps, long_repeats=[round(x, 4) for x in rep_long], long_min=round(min(rep_long), 4),
                  long_median=round(sorted(rep_long)[len(rep_long) // 2], 4), long_max=round(max(rep_long), 4))
    del wl
    train = None
    if not args.no_train:
        net._engines.clear()
        torch.cuda.empty_cache()
        try:
            train = train_bench(args.cfg, args.img_size, args.train_batch, args.train_steps, 2, world, local_rank, device, barrier)
        except Exception as e:        # the headline line must still be printed; the failure is reported under extra.train
            train = dict(error=f'{type(e).__name__}: {e}'[:400])
    train16 = None
    config4 = world == 8 or os.environ.get('YM_BENCH_CONFIG4', '0') == '1'          # (the env switch lets a 1-GPU test walk this path)
    if not args.no_train and config4 and args.train_batch != 16 and 'error' not in (train or {}):
        # BASELINE config 4: bs=16 per GPU on the full node (every rank takes part: the step has collectives)
        try:
            torch.cuda.empty_cache()
            train16 = train_bench(args.cfg, args.img_size, 16, 4, 2, world, local_rank, device, barrier)
        except Exception as e:
            train16 = dict(error=f'{type(e).__name__}: {e}'[:400])

    out = None
    if rank == 0:
        # forward-only rate on the same engine, and the live conv roofline
        fw = Workload(net, cfg, args.batch, args.img_size, device, with_post=False)
        k1 = max(args.steps, 50)
        t_fwd = min(timed(fw, k1, 2, lambda: None), timed(fw, k1, 0, lambda: None)) / k1      # ONE request at a time (graph replay)
        t_fwd_multi, single, latency = t_fwd, None, None
        if inflight > 1:
            fwm = Workload(net, cfg, args.batch, args.img_size, device, with_post=False, inflight=inflight)
            t_fwd_multi = timed(fwm, 4 * k1, 4, lambda: None) / (4 * k1)      # seconds per image, requests overlapped
            del fwm
            one = Workload(net, cfg, args.batch, args.img_size, device, with_post=not args.no_post, inflight=1)
            t_one = min(timed(one, k1, 5, lambda: None), timed(one, k1, 0, lambda: None)) / k1
            del one
            # per-request device latency (HIP events on the slot's stream around forward + nms + after_nms) with `inflight` requests
            # overlapped, and the Little's-law check: requests in flight = throughput x latency must come out as `inflight` if the
            # requests really overlap (one at a time it would be 1)
            lat = Workload(net, cfg, args.batch, args.img_size, device, with_post=not args.no_post, inflight=inflight, timed=True)
            n_lat = (400 if not args.lean else 16) // max(1, args.batch) or 2
            t_lat = timed(lat, n_lat, 8, lambda: None) / n_lat
            ls = sorted(lat.pipe.latencies_ms[-n_lat:])
            del lat
            mean_l = sum(ls) / len(ls)
            latency = dict(unit='ms', requests=len(ls), p50=round(ls[len(ls) // 2], 3), p99=round(ls[min(len(ls) - 1, int(len(ls) * 0.99))], 3),
                           mean=round(mean_l, 3), min=round(ls[0], 3), max=round(ls[-1], 3), ms_per_step=round(t_lat * 1e3, 4),
                           requests_in_flight_by_littles_law=round(mean_l / (t_lat * 1e3), 3),
                           single_request_ms=round(t_one * 1e3, 3),
                           note='device time of ONE request (its own stream: forward + nms + after_nms) while the others run; '
                                'throughput x latency = requests in flight')
        flops, conv_secs, launches, layers = conv_roofline(fw.engine, fw.img)
        achieved = flops / conv_secs / 1e12
        traffic, traffic_src = None, None
        pmc_path = next((q for q in (os.path.join(REPO, 'profiles', f'r0{r}_pmc_hbm_infer_bs1_res101.json') for r in (6, 5, 4, 3))
                         if os.path.exists(q)), '')
        if args.cfg == 'res101_coco' and args.batch == 1 and pmc_path:
            # HBM-side bytes per conv launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE x2
            # gfx950 correction + WRITE_SIZE); not re-measured live (bench.py cannot wrap itself in rocprofv3)
            traffic = round(json.load(open(pmc_path))['conv_kernels']['traffic_bytes_per_launch'])
            traffic_src = os.path.relpath(pmc_path, REPO)
        # `achieved`: the conv kernels' algorithmic flops over the GRAPH-REPLAY forward (the path `value` times; its few non-conv
        # kernels — layout, max-pool, 3 upsamples, softmax: ~2 % — are left in the denominator, so this is a lower bound that agrees
        # with the rocprofv3 kernel trace under profiles/).  The eager per-launch HIP-event figure is kept beside it.
        achieved_graph = flops / t_fwd / 1e12
        single = dict(achieved=round(achieved_graph, 2), frac=round(achieved_graph / F32_MFMA_PEAK_TFLOPS, 4),
                      avg_launch_us=round(t_fwd / launches * 1e6, 2), forward_graph_ms=round(t_fwd * 1e3, 3),
                      eager_event_conv_ms=round(conv_secs * 1e3, 3), eager_event_frac=round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                      note='one request at a time: the per-launch figure (algorithmic flops of a conv launch / its duration in the dependent chain)')
        if inflight > 1:
            single['img_s_with_post'] = round(args.batch / t_one, 2)
        # with `inflight` requests overlapped the conv launches of different requests share the chip, so a launch's own duration no
        # longer measures anything; `achieved` is then the conv kernels' algorithmic flops of the timed forwards / their wall time
        achieved_multi = flops / t_fwd_multi / 1e12
        # `achieved` / `frac`: the conv kernels' algorithmic flops executed inside the SAME timed region as `value` (K steps, forward
        # + nms + after_nms, `inflight` requests overlapped) / that region's wall time: the post-processing and the few non-conv
        # kernels stay in the denominator, so this is a lower bound of the conv kernels' own rate.  With requests overlapped a
        # launch's own duration measures nothing (launches of different requests share the chip): the per-launch figure is
        # `single_request` (one request at a time), the forward-only aggregate is `forward_only`.
        achieved_region = region_flops / elapsed_local / 1e12
        roofline = dict(bound='mfma', achieved=round(achieved_region, 2), peak=F32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                        frac=round(achieved_region / F32_MFMA_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                        kernel='conv_igemm_f32 / conv_igemm_pers / conv_wdma_f32 (all instantiations)', launches_per_step=launches,
                        flops_per_launch=round(flops / launches), avg_launch_us=round(elapsed_local / args.steps / launches * 1e6, 2),
                        requests_in_flight=inflight,
                        definition=('algorithmic conv flops of the timed steps / wall time of the SAME region that `value` times '
                                    '(forward + nms + after_nms, requests_in_flight requests overlapped)'),
                        forward_only=dict(achieved=round(achieved_multi, 2), frac=round(achieved_multi / F32_MFMA_PEAK_TFLOPS, 4),
                                          forward_graph_ms=round(t_fwd_multi * 1e3, 3),
                                          note='separate region, forwards only, same streams'),
                        single_request=single)
        extra = dict(spread=spread, latency=latency, forward_only_ms=round(t_fwd_multi * 1e3, 3),
                     forward_only_img_s=round(args.batch / t_fwd_multi, 1),
                     forward_tflops=round(flops / t_fwd_multi / 1e12, 2),
                     gflop_per_img=round(flops / args.batch / 1e9, 1))
        slow = sorted(layers, key=lambda l: -l['ms'])[:5]
        extra['slowest_convs'] = [dict(name=l['name'], ms=round(l['ms'], 4), tflops=round(l['gflop'] / l['ms'], 1)) for l in slow]
        if not args.no_extra and world == 1:
            for name, b in ((args.cfg, 8), ('res50_coco', 8), ('swin_tiny_coco', 8)):
                # bs=8 (BASELINE config 2 / 5): one batch at a time, and with 2 batches in flight (the same RequestPipeline)
                n2, c2 = (net, cfg) if name == args.cfg else build_net(name, args.img_size, device)
                w2 = Workload(n2, c2, b, args.img_size, device, with_post=not args.no_post)
                t2 = timed(w2, 8, 2, lambda: None)
                f2 = Workload(n2, c2, b, args.img_size, device, with_post=False)
                tf2 = timed(f2, 8, 2, lambda: None) / 8
                fl2 = f2.engine.total_flops
                del w2, f2
                w3 = Workload(n2, c2, b, args.img_size, device, with_post=not args.no_post, inflight=2)
                t3 = timed(w3, 16, 4, lambda: None) / 16
                del w3
                f3 = Workload(n2, c2, b, args.img_size, device, with_post=False, inflight=2)
                tf3 = timed(f3, 16, 4, lambda: None) / 16
                del f3
                tf4 = None
                if name == args.cfg:                  # four batches in flight (forward only), the named backbone only
                    f4 = Workload(n2, c2, b, args.img_size, device, with_post=False, inflight=4)
                    tf4 = timed(f4, 24, 8, lambda: None) / 24
                    del f4
                if name.startswith('swin'):
                    fl2 += 1.8e9 * b          # attention matmuls (QK^T, PV), not run by the conv kernel
                extra[f'{name}_bs{b}'] = dict(img_s=round(b / t3, 1), forward_only_img_s=round(b / tf3, 1),
                                              forward_tflops=round(fl2 / tf3 / 1e12, 2),
                                              frac_f32_mfma_peak=round(fl2 / tf3 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4), batches_in_flight=2,
                                              one_batch_at_a_time=dict(img_s=round(b * 8 / t2, 1), forward_only_img_s=round(b / tf2, 1),
                                                                       frac_f32_mfma_peak=round(fl2 / tf2 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)))
                if tf4 is not None:
                    extra[f'{name}_bs{b}']['four_batches_in_flight'] = dict(
                        forward_only_img_s=round(b / tf4, 1), frac_f32_mfma_peak=round(fl2 / tf4 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4))
        if not args.no_extra and world == 1:
            extra['post'] = post_bench(net, cfg, device, args.img_size)
            if args.batch == 1 and not args.no_post:
                extra['chained'] = chained_bench(args.cfg, args.img_size, device, inflight)
            # split-bf16 fast modes (ym_conv_desc.mma): same plan, same fp32 tensors, products on the bf16 MFMA.  bf16x3 holds the
            # reference's 544 px goldens inside the 1e-4 bar (tests/test_gpu_forward.py::test_forward_544_bs8_split_bf16_modes_...);
            # `value` above stays the f32 parity mode.  Roofline here: 3 (6) bf16 MFMA flops per algorithmic flop vs 2.5 PF dense.
            split = {}
            for name, b, mma in ((args.cfg, 1, 3), (args.cfg, 8, 3), ('res50_coco', 8, 3), ('swin_tiny_coco', 8, 3)):
                n2, c2 = (net, cfg) if name == args.cfg else build_net(name, args.img_size, device)
                w2 = Workload(n2, c2, b, args.img_size, device, with_post=not args.no_post)
                w2.engine.set_mma(mma)
                k2 = 40 if b == 1 else 10
                t2 = timed(w2, k2, 5, lambda: None) / k2
                f2 = Workload(n2, c2, b, args.img_size, device, with_post=False)
                tf2 = min(timed(f2, k2, 5, lambda: None), timed(f2, k2, 2, lambda: None)) / k2      # (best of two: a short region)
                fl2 = f2.engine.total_flops
                split[f'{name}_bs{b}_bf16x{mma}'] = dict(img_s=round(b / t2, 1), forward_only_img_s=round(b / tf2, 1),
                                                         tflops_f32_equiv=round(fl2 / tf2 / 1e12, 1),
                                                         frac_bf16_mfma_peak=round(mma * fl2 / tf2 / 1e12 / 2500.0, 4),
                                                         convs_on_bf16_mfma=sum(1 for c in f2.engine.convs if c.mma))
                w2.engine.set_mma(0)
            extra['split_bf16'] = split
            extra['eval_metrics'] = eval_metrics_bench(device, cpu=not args.no_cpu_baseline)
            if args.batch == 1:
                try:
                    extra['other_sizes'] = other_sizes_bench(args.cfg, device)
                except Exception as e:
                    extra['other_sizes'] = dict(error=f'{type(e).__name__}: {e}'[:400])
            if args.batch == 1 and not args.no_post:
                try:
                    # a process of its own, as `dropin/run.py eval.py` is one (GPU_MAX_HW_QUEUES=8 like the launcher exports for eval.py):
                    # inside this process, after the pipelines / engines of the other legs, one of the loop's legs picks up ~2 ms of
                    # host-side allocator / wait time per image, a different one from run to run (283 vs 171 img/s for the same loop)
                    import subprocess
                    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
                    env['GPU_MAX_HW_QUEUES'] = '8'
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', 'eval_loop', '--cfg', args.cfg, '--img_size',
                                        str(args.img_size)], env=env, capture_output=True, text=True, timeout=900)
                    extra['eval_loop'] = json.loads(r.stdout.strip().splitlines()[-1])
                    extra['eval_loop']['process'] = 'own (as dropin/run.py eval.py)'
                except Exception as e:
                    extra['eval_loop'] = dict(error=f'{type(e).__name__}: {e}'[:400])
            extra['train_aug'] = train_aug_bench(device, cpu=not args.no_cpu_baseline)
            extra['ann_to_mask'] = ann_to_mask_bench(device, cpu=not args.no_cpu_baseline)
            if not args.no_train and args.cfg != 'swin_tiny_coco':
                net._engines.clear()
                torch.cuda.empty_cache()
                try:
                    # a process of its own: this one started HIP with GPU_MAX_HW_QUEUES=8 for the request pipeline, and the reference's
                    # training loop (torch DDP + RCCL + side stream) wants the default 4 (dropin/run.py::hw_queues_for)
                    import subprocess
                    env = {k: v for k, v in os.environ.items() if k not in ('GPU_MAX_HW_QUEUES', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
                    env['GPU_MAX_HW_QUEUES'] = '4'
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', 'train_reference_loop', '--cfg', args.cfg, '--img_size',
                                        str(args.img_size), '--train-batch', str(args.train_batch), '--train-steps', str(max(args.train_steps, 10))],
                                       env=env, capture_output=True, text=True, timeout=900)
                    extra['train_reference_loop'] = json.loads(r.stdout.strip().splitlines()[-1])
                    extra['train_reference_loop']['GPU_MAX_HW_QUEUES'] = 4
                except Exception as e:
                    extra['train_reference_loop'] = dict(error=f'{type(e).__name__}: {e}'[:400])
                torch.cuda.empty_cache()
                extra['train_swin_tiny_coco'] = train_bench('swin_tiny_coco', args.img_size, args.train_batch, args.train_steps, 2, 1,
                                                            local_rank, device, lambda: None)
                # opt-in fast training mode: forward + data-gradient convs on the bf16 MFMA (bf16x3 split products; NOT parity-grade,
                # weight gradients stay on the f32 MFMA) and config 4's per-GPU batch (16) in the f32 parity mode
                os.environ['YM_TRAIN_MMA'] = '3'
                try:
                    extra['train_bf16x3_fwd_dgrad'] = train_bench(args.cfg, args.img_size, args.train_batch, args.train_steps, 2, 1, local_rank,
                                                                  device, lambda: None)
                finally:
                    os.environ['YM_TRAIN_MMA'] = '0'
                torch.cuda.empty_cache()
                extra['train_bs16'] = train_bench(args.cfg, args.img_size, 16, 4, 2, 1, local_rank, device, lambda: None)
                try:
                    import subprocess
                    env = {k: v for k, v in os.environ.items() if k not in ('GPU_MAX_HW_QUEUES', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', 'train_to_map'], env=env, capture_output=True,
                                       text=True, timeout=600)
                    extra['train_to_map'] = json.loads(r.stdout.strip().splitlines()[-1])
                except Exception as e:
                    extra['train_to_map'] = dict(error=f'{type(e).__name__}: {e}'[:400])
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the oracle's throughput depends on the thread count (128 threads on a 256-cpu shared host are SLOWER than 8: the
            # layers are too small to split that far), so the sample is taken at 8 / 32 / all threads and the best one is `value`;
            # 8 threads is also SURVEY §8d's comparison point with the 8-vCPU survey probes.  BASELINE.json config 1
            # (res50_coco bs=1 on the CPU path) is measured the same way.
            runs = {t: cpu_baseline(args.cfg, args.img_size, threads=t, budget_s=4.0, max_img=4) for t in (8, 32, None)}
            cpu = dict(max(runs.values(), key=lambda r: r['value']))
            cpu['by_threads'] = {str(r['cores']): r['value'] for r in runs.values()}
            r50 = [cpu_baseline('res50_coco', args.img_size, threads=t, budget_s=4.0, max_img=4) for t in (8, 32)]
            cpu['res50_coco'] = max(r50, key=lambda r: r['value'])
        extra['train'] = train
        if train16 is not None:
            extra['train_bs16_per_gpu_ddp8'] = train16
        if train is not None and 'ddp' in train:          # N > 1: the north star's DDP curve is THIS block (`value` stays the replicas)
            extra['ddp'] = train['ddp']
        primary_train = args.mode == 'train' and train is not None and 'error' not in train
        out = {
            'metric': (f'img/s {args.cfg} 544x544 DDP training (bs={args.train_batch}/GPU)' if primary_train else
                       f'img/s {args.cfg} 544x544 inference (bs={args.batch}/GPU'
                       + (f', {inflight} requests in flight)' if inflight > 1 else ')')),
            'value': train['img_s'] if primary_train else round(value, 2), 'unit': 'img/s', 'n_gpus': world,
            # the reference's own loop (eval.py:36-69: one image at a time) on the same GPU, beside the overlapped figure
            'value_single_request': single.get('img_s_with_post') if single else None,
            'steps': train['steps'] if primary_train else args.steps, 'warmup': train['warmup'] if primary_train else args.warmup,
            'ms_per_step': train['ms_per_step'] if primary_train else round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{args.cfg} 544x544 (the reference\'s "550-class" size) bs={args.batch} inference: '
                                   f'forward + nms + after_nms(480x640) per image'
                                   + (f', {inflight} independent bs={args.batch} requests in flight on {inflight} HIP streams' if inflight > 1 else '')
                                   if not args.no_post else
                                   f'{args.cfg} 544x544 bs={args.batch} forward only',
                       'global_batch': args.batch * world, 'parallelism': f'replicas x{world} (inference does not shard)',
                       'weights': 'seeded random init', 'post_inputs': 'synthetic dense head outputs (17.8k candidates)',
                       'requests_in_flight': inflight},
            'roofline': roofline, 'cpu_baseline': cpu, 'extra': extra,
        }
    barrier()
    if rank == 0:
        os.write(record_fd, (json.dumps(out) + '\n').encode())
    if dist is not None:
        torch.cuda.synchronize()          # nothing in flight when the process group (and its watchdog thread) is torn down
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
