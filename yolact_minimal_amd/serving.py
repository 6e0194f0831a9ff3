"""bs=1 serving pipeline: the forward of image i+1 overlaps the post-processing of image i.

The reference's loop (`eval.py:40-52`, `detect.py:60-80`) is strictly sequential per image: forward, `nms` (which ends in a
4-byte host read of the detection count), `after_nms`.  At bs=1 the forward is a chain of ~140 small launches that leaves CUs
idle in every tail, and the post-processing is a handful of tiny kernels plus that host round trip — the two overlap well.
`ServingPipeline` keeps the engine's hipGraph on one HIP stream, stages the four prediction tensors into one of two slots
(8 MB of device copies) and runs `nms` + `after_nms` of that slot on a second stream while the next forward is already in
flight; events guard the slots, there is no extra host synchronisation.  Results are identical to the sequential calls.

Measured on one MI355X (res101_coco 544, 17.8 k candidates): 304 img/s pipelined vs 311 img/s sequential — the forward's
kernels already occupy every CU and the two streams do not overlap usefully at this kernel granularity, so `bench.py` keeps
the sequential order as its default (`--pipeline` selects this class).  It is kept for callers whose post-processing is
heavier (larger `after_nms` targets, many detections) or who need the staging / two-slot structure for asynchronous I/O.
"""
import torch

from .utils.output_utils import nms, after_nms


class ServingPipeline:
    def __init__(self, net, cfg, img_size, device):
        self.net, self.cfg, self.device = net, cfg, device
        example = torch.zeros(1, 3, img_size, img_size, device=device)
        self.engine = net._engine(example)
        self.engine.run(example)                      # builds / captures the plan on the current stream
        torch.cuda.synchronize(device)
        self.s_fwd, self.s_post = torch.cuda.Stream(device), torch.cuda.Stream(device)
        self.stage = [tuple(torch.empty_like(t) for t in self.engine.outputs()) for _ in range(2)]
        self.ev_done = [torch.cuda.Event() for _ in range(2)]
        self.ev_free = [torch.cuda.Event() for _ in range(2)]
        self.ev_out = torch.cuda.Event()
        self.submitted = self.collected = 0

    def submit(self, img):
        """Enqueue the forward of one image [1, 3, S, S] (asynchronous)."""
        if self.submitted - self.collected >= 2:
            raise RuntimeError('ServingPipeline: two images are already in flight; collect() one first')
        slot = self.submitted % 2
        self.s_fwd.wait_stream(torch.cuda.current_stream(self.device))      # the caller produced `img` on its stream
        with torch.cuda.stream(self.s_fwd):
            if self.submitted >= 2:
                self.s_fwd.wait_event(self.ev_free[slot])                   # post-processing of image k-2 is done with the slot
            self.engine.run(img)
            for dst, src in zip(self.stage[slot], self.engine.outputs()):
                dst.copy_(src, non_blocking=True)
            self.ev_done[slot].record(self.s_fwd)
        img.record_stream(self.s_fwd)
        self.submitted += 1

    def collect(self, img_h, img_w, post_inputs=None):
        """`nms` + `after_nms` of the oldest image in flight -> (ids, scores, boxes, masks) or 4 x None, like the reference.
        `post_inputs` (bench.py only) substitutes synthetic head outputs for the staged predictions of a random-init net."""
        if self.collected >= self.submitted:
            raise RuntimeError('ServingPipeline: nothing in flight')
        slot = self.collected % 2
        with torch.cuda.stream(self.s_post):
            self.s_post.wait_event(self.ev_done[slot])
            cls, box, coef, proto = self.stage[slot] if post_inputs is None else post_inputs
            r = nms(cls, box, coef, proto, self.net.anchors, self.cfg)        # one 4-byte host read, on this stream only
            self.ev_free[slot].record(self.s_post)                            # nms has gathered what it needs from the slot ...
            out = after_nms(r[0], r[1], r[2], r[3], r[4], img_h, img_w, self.cfg)
            if r[4] is not None:
                self.ev_free[slot].record(self.s_post)                        # ... except proto (a view of it): release after after_nms
            self.ev_out.record(self.s_post)
        torch.cuda.current_stream(self.device).wait_event(self.ev_out)        # results are safe to use on the caller's stream
        for t in out:
            if torch.is_tensor(t):
                t.record_stream(torch.cuda.current_stream(self.device))
        self.collected += 1
        return out

    def run(self, images, sizes):
        """Generator over (ids, scores, boxes, masks) for an iterable of images with their original (h, w)."""
        images, sizes = list(images), list(sizes)
        if images:
            self.submit(images[0])
        for i in range(len(images)):
            if i + 1 < len(images):
                self.submit(images[i + 1])
            yield self.collect(*sizes[i])
