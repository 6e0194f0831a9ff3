"""Named wall-clock sections with device fences — the module-level API the reference's harness scripts use
(`/root/reference/utils/timer.py`: `reset`, `start`, `counter(name)`, `add_batch_time`, `get_times`; call sites
`eval.py:33-80`, `train.py:88-177`, `detect.py:59-155`).  Host bookkeeping only.

Semantics kept: nothing is recorded before `start()` (the first iteration is excluded by the callers); every section is fenced
with a device synchronize on entry and exit (the kernels are asynchronous); each series is a sliding window of `length`
samples; `data` = batch time minus the latest sample of every other section."""
import time
from collections import deque

import torch


class _Registry:
    def __init__(self, length=100):
        self.length = length
        self.running = False
        self.series = {'batch': deque(maxlen=length), 'data': deque(maxlen=length)}

    def window(self, name):
        if name not in self.series:
            self.series[name] = deque(maxlen=self.length)
        return self.series[name]


_reg = _Registry()


def _fence():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def reset(length=100):
    global _reg
    _reg = _Registry(length)


def start():
    if any(len(v) for v in _reg.series.values()):
        print('Warning, time list is not empty when starting.')
    _reg.running = True


def add_batch_time(batch_time):
    if not _reg.running:
        return
    inner = sum(v[-1] for k, v in _reg.series.items() if k not in ('batch', 'data') and len(v))
    _reg.series['batch'].append(batch_time)
    _reg.series['data'].append(batch_time - inner)


def get_times(names):
    return [sum(_reg.series[n]) / len(_reg.series[n]) if len(_reg.series.get(n, ())) else float('nan') for n in names]


class counter:
    """`with timer.counter('forward'): ...` — one sample of section `name` (skipped until `start()`)."""

    def __init__(self, name, trt_mode=False):
        self.name, self.fenced, self.reg = name, not trt_mode, _reg
        self.active = _reg.running

    def __enter__(self):
        if self.active:
            if self.fenced:
                _fence()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.active:
            if self.fenced:
                _fence()
            self.reg.window(self.name).append(time.perf_counter() - self.t0)
        return False
