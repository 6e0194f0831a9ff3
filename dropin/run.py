#!/usr/bin/env python3
"""Launcher: run one of the reference's scripts, unmodified, on the MI355X path.

    cd /path/to/Yolact_minimal
    python /path/to/repo/dropin/run.py eval.py --weight weights/best_30.5_res101_coco_392000.pth
    python -m torch.distributed.run --nproc-per-node 8 /path/to/repo/dropin/run.py train.py --train_bs 64

Python puts the SCRIPT's directory first on the module path, so a checkout's own `modules/`, `utils/`, `config.py` would win
over PYTHONPATH.  This launcher orders the path as [dropin, repo, checkout, ...] and then executes the script as `__main__`
(with `sys.argv` shifted), so `from modules.yolact import Yolact`, `from utils.output_utils import nms, after_nms`,
`from config import get_config` resolve to yolact_minimal_amd while everything else still comes from the checkout."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def hw_queues_for(script):
    """ROCm multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the runtime
    starts, i.e. before the script's `import torch` touches HIP.  Serving (eval.py / detect.py: requests in flight on separate
    streams, yolact_minimal_amd.pipeline.RequestPipeline) wants 8 — two request streams that share a queue do not overlap.  Training
    wants the default: the part has four compute pipes, and torch DDP + RCCL + the weight-gradient side stream keep more than four
    queues busy once they each have one; measured on res101 bs=8 with the timer fences removed: 54.7 ms per step with 8 queues, 42.7
    with 4 (profiles/r06_reference_loop_timings.txt).  An exported GPU_MAX_HW_QUEUES always wins."""
    name = os.path.basename(script)
    return '8' if name.startswith(('eval', 'detect')) else None


def host_gc_policy():
    """The reference's loops build a few hundred small containers per image (eval.py:59-67: two record dicts per detection kept in
    `MakeJson`; `APDataObject` points; per-step loss lists in train.py).  CPython's generational collector answers a growing set of
    survivors with FULL collections, each one a walk over the whole heap — mostly torch's import-time objects — which at 3 ms of
    device work per image is no longer noise: the `--coco_api` loop with device RLE ran at 211-246 img/s with the default policy
    and 344 with the collector off (`YM_GC_OFF=1 python bench.py --leg eval_loop`).  So, once the imports are done: collect, move
    what exists to the permanent generation (`gc.freeze`: never scanned again) and let the young generation fill further before a
    pass.  Nothing is leaked or disabled; `YM_DROPIN_GC=0` keeps the interpreter's defaults."""
    if os.environ.get('YM_DROPIN_GC', '1') == '0':
        return
    import gc
    import torch  # noqa: F401  (the heap worth freezing; GPU_MAX_HW_QUEUES is exported before this line)
    import yolact_minimal_amd.modules.yolact  # noqa: F401
    gc.collect()
    gc.freeze()
    gc.set_threshold(20000, 20, 20)


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    q = hw_queues_for(script)
    if q is not None:
        os.environ.setdefault('GPU_MAX_HW_QUEUES', q)
    checkout = os.path.dirname(script)
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, REPO, checkout)]
    sys.path[:] = [HERE, REPO, checkout] + rest
    sys.argv = [script] + sys.argv[2:]
    host_gc_policy()
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
