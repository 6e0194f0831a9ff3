"""Serving with several independent requests in flight on one MI355X (batch 1: 400 -> 600-617 img/s; batch 8: 700 -> 760 forward-only).

The reference evaluates one image at a time (`eval.py:36-69`: forward -> nms -> after_nms, a device synchronisation around each).
On this part a bs=1 forward is a chain of ~190 dependent launches of 0.6-1.4 GFLOP each: every launch pays a kernel boundary, an
address set-up + cold-L2 round trip and an epilogue (~9 us in round 3, ~6 us since round 4) around ~7 us of MFMA work, so ONE chain keeps the matrix pipe ~35-43 %
busy whatever the kernels do.  Requests are independent, so the way to fill the chip at batch size 1 is to have several of them in
flight: `RequestPipeline` owns `depth` complete engines (activations, split-K scratch, arrival counters, hipGraph: nothing shared
but the read-only weights) and `depth` HIP streams, and runs request i on slot i % depth.  Each request is still ONE image through
`Yolact.forward` -> `nms` -> `after_nms` with ONE host read (its detection count, which `after_nms` needs to size what it returns);
the count is copied to pinned host memory behind the request and read when the slot comes up again, so the host never waits on
the request it has just enqueued.

Measured mid-round 3 (res101_coco 544 px, MI355X, forward + nms + after_nms(480x640)): depth 1: 325 img/s, 2: 468, 3: 545, 4: 594, 5: 495,
8: 479; with round 4's kernels 3: 580, 4: 603-617, 5: 499, 6: 527 -- the part schedules four compute pipes; GPU_MAX_HW_QUEUES must be >= depth + 1 (ROCm multiplexes HIP streams onto 4
hardware queues by default and two streams that share a queue do not overlap): set it to 8 before the first HIP call
(`RequestPipeline` warns when the variable is missing or too small: it cannot be changed once the runtime is up).
"""
import os
import warnings

import torch

from .engine import InferEngine
from .utils.output_utils import nms_batch, after_nms_batch

_streams = {}


def _stream_set(device, n):
    """One process-wide set of streams per device: every pipeline overlaps on the SAME streams (= the same hardware queues)."""
    key = torch.device(device)
    lst = _streams.setdefault(key, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst[:n]


def hw_queues_ok(depth):
    """Does the HIP runtime have a hardware queue for each of `depth` slot streams plus the caller's stream?  (ROCm reads
    GPU_MAX_HW_QUEUES when it creates its first stream; the default is 4.)"""
    try:
        q = int(os.environ.get('GPU_MAX_HW_QUEUES', '4'))
    except ValueError:
        q = 4
    return q >= depth + 1


class RequestPipeline:
    """`depth` requests in flight.  `submit()` returns the FINISHED result of the request that used the slot before; results are
    fresh tensors the caller owns (network outputs are copied off the slot's buffers before the slot runs again).
    `return_outputs=False` (throughput measurements of the forward alone): `submit()` / `drain()` hand back nothing for
    `with_post=False` and the copy is skipped.  `timed=True`: every request is bracketed by HIP events on its slot's stream and
    `latencies_ms` collects the per-request device latency (bench.py: p50 / p99 and the Little's-law check)."""

    def __init__(self, net, cfg, height, width, device, depth=4, out_hw=(480, 640), with_post=True, batch=1, return_outputs=True,
                 timed=False):
        self.net, self.cfg, self.device, self.depth, self.batch = net, cfg, torch.device(device), depth, batch
        self.out_hw, self.with_post, self.return_outputs, self.timed = out_hw, with_post, return_outputs, timed
        if depth > 1 and not hw_queues_ok(depth):
            warnings.warn(f'RequestPipeline(depth={depth}): GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)")} < '
                          f'{depth + 1}; HIP streams that share a hardware queue do not overlap -- export GPU_MAX_HW_QUEUES=8 before the '
                          f'first HIP call', RuntimeWarning, stacklevel=2)
        # slots of a pipeline with requests in flight read the throughput-tuned entries of the table (engine._entry)
        self.engines = [InferEngine(net, batch, height, width, device, mode='throughput' if depth > 1 else 'latency') for _ in range(depth)]
        self.streams = _stream_set(device, depth)
        self.counts_host = [torch.zeros(batch, dtype=torch.int32).pin_memory() for _ in range(depth)]
        self.events = [torch.cuda.Event() for _ in range(depth)]
        self.t0 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)] if timed else None
        self.t1 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)] if timed else None
        self.latencies_ms = []
        self.pending = [None] * depth
        self.anchors = torch.tensor(net.anchors, dtype=torch.float32).reshape(-1, 4).to(device) \
            if not torch.is_tensor(net.anchors) else net.anchors.to(device)
        self.vt = float(getattr(cfg, 'visual_thre', 0) or 0)
        self.submitted = 0
        self.detections = 0

    def warm_up(self, img, head_outputs=None, rounds=4):
        """Capture every slot's hipGraph (first run of an engine) and push `rounds` requests through every slot, outside any timed
        region: the first requests of a pipeline allocate their result tensors (123 MB of masks per request at 480x640) with
        hipMalloc until the caching allocator holds enough blocks for `depth` requests in flight plus the results the caller
        still owns, and the part ramps its clocks; a server is warm, so the pipeline warms itself.

        Side effects the caller should know: `submitted`, `detections` and `latencies_ms` are RESET to zero / empty afterwards
        (statistics collected before the call are discarded), and with the default `head_outputs=None` the warm-up requests
        post-process the network's own outputs (a random-init net: degenerate detection counts, full-size mask tensors all the
        same).  `rounds=0` captures the graphs only (the cost of the pipeline before round 5)."""
        torch.cuda.synchronize(self.device)
        for e, st in zip(self.engines, self.streams):
            with torch.cuda.stream(st):
                e.run(img)
        torch.cuda.synchronize(self.device)
        if rounds <= 0:
            return
        timed, self.timed = self.timed, False
        for _ in range(rounds * self.depth):
            self.submit(img, head_outputs)
        self.drain()
        torch.cuda.synchronize(self.device)
        self.timed = timed
        self.submitted = self.detections = 0
        self.latencies_ms = []

    def finish(self, slot):
        """Result of the request that last used `slot`: (ids, scores, boxes_px, masks) like `after_nms` (None x 4 without
        detections -- also when `cfg.visual_thre` removes all of them, `utils/output_utils.py:204-212` of the reference; a list of
        such tuples, one per image, when batch > 1), or None if the slot is idle.  Without post-processing: copies of the slot's
        four network outputs (None with `return_outputs=False`)."""
        pend = self.pending[slot]
        if pend is None:
            return None
        self.pending[slot] = None
        if not self.with_post:
            pend.synchronize()
            if self.timed:
                self.latencies_ms.append(self.t0[slot].elapsed_time(self.t1[slot]))
            if not self.return_outputs:
                return None
            # the slot's own buffers are overwritten by its next request: hand out copies, made on the caller's stream (the slot's
            # next run is ordered behind that stream by `submit`)
            return tuple(t.clone() for t in self.engines[slot].outputs())
        ids, scores, box_px, masks, counts, ev = pend
        ev.synchronize()                                  # THIS request only
        if self.timed:
            self.latencies_ms.append(self.t0[slot].elapsed_time(self.t1[slot]))
        # the results were allocated on the slot's stream and are consumed on the caller's: tell the allocator, or the slot's next
        # request could be handed the block while a kernel of the caller still reads it
        cur = torch.cuda.current_stream(self.device)
        for t in (ids, scores, box_px, masks):
            t.record_stream(cur)
        out = []
        for b, n in enumerate(self.counts_host[slot].tolist()):
            if n == 0:
                out.append((None, None, None, None))
                continue
            r = (ids[b, :n], scores[b, :n], box_px[b, :n], masks[b, :n])
            if self.vt > 0:                               # detect.py's score filter, as in `after_nms` / `after_nms_batch(sync=True)`
                keep = r[1] >= self.vt
                r = tuple(t[keep] for t in r) if bool(keep.any()) else (None, None, None, None)
            if r[0] is not None:
                self.detections += int(r[0].shape[0])
            out.append(r)
        return out[0] if self.batch == 1 else out

    def submit(self, img, head_outputs=None):
        """Enqueue one request ([batch,3,H,W] images, device resident) on the next slot and return the finished result of the request
        that used this slot before (None the first `depth` times).  `head_outputs`: post-process these (class, box, coef, proto)
        tensors instead of the forward's own outputs (bench.py: a random-init network yields degenerate detections).
        The request is ordered behind everything the caller's current stream has queued; the caller must not overwrite `img` before
        the request has finished (use one input buffer per slot when images arrive by H2D copy)."""
        slot = self.submitted % self.depth
        self.submitted += 1
        done = self.finish(slot)
        ev = self.events[slot]
        # `img` / `head_outputs` were produced on the caller's stream (an H2D copy, `val_aug`), and `finish` may have queued copies
        # of the slot's previous outputs there: the slot's stream starts behind that
        self.streams[slot].wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.streams[slot]):
            eng = self.engines[slot]
            if self.timed:
                self.t0[slot].record()
            eng.run(img)
            if self.with_post:
                cls, box, coef, proto = head_outputs if head_outputs is not None else eng.outputs()
                r = after_nms_batch(nms_batch(cls, box, coef, proto, self.anchors, self.cfg), self.out_hw[0], self.out_hw[1], self.cfg,
                                    sync=False)
                self.counts_host[slot].copy_(r[4], non_blocking=True)
                if self.timed:
                    self.t1[slot].record()
                ev.record()
                self.pending[slot] = r + (ev,)
            else:
                if self.timed:
                    self.t1[slot].record()
                ev.record()
                self.pending[slot] = ev
        return done

    def drain(self):
        """Finish everything in flight, oldest first."""
        out = []
        for k in range(self.depth):
            slot = (self.submitted + k) % self.depth
            busy = self.pending[slot] is not None
            r = self.finish(slot)
            if busy and (r is not None or self.with_post):
                out.append(r)
        return out
