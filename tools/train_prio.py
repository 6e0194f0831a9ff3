"""Does the hardware-queue priority of the two training streams matter?  One training measurement per process:

    YM_WGRAD_STREAM_PRIORITY=-1 python tools/train_prio.py          # weight gradients (side stream) on a high-priority queue
    MAIN_PRIO=-1 python tools/train_prio.py                          # the step itself (forward / loss / data gradients) on one
    python tools/train_prio.py                                       # both on normal queues (the product's default)

Prints ms per step of res101_coco 544 px bs=8 (bench.py's `extra.train` workload, 2 warm-up + 8 timed steps, best of 2 regions).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import torch                                                                     # noqa: E402

import bench                                                                     # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    prio = os.environ.get('MAIN_PRIO')
    ctx = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=int(prio))) if prio else torch.cuda.stream(torch.cuda.current_stream(dev))
    best = None
    with ctx:
        for _ in range(2):
            r = bench.train_bench('res101_coco', 544, batch, 8, 2, 1, 0, dev, lambda: None)
            best = r if best is None or r['ms_per_step'] < best['ms_per_step'] else best
    print(f"main_prio={prio or 0} side_prio={os.environ.get('YM_WGRAD_STREAM_PRIORITY', '0')} streams={os.environ.get('YM_WGRAD_STREAMS', '1')} "
          f"bs={batch}: {best['ms_per_step']} ms/step  finite={best['finite']} losses={best['last_losses']}")


if __name__ == '__main__':
    main()
