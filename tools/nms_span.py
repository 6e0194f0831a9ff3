#!/usr/bin/env python3
"""Device-side anatomy of one `nms` call from a rocprofv3 --kernel-trace run of tools/post_time.py (rocpd sqlite):
   nms_span.py results.db
For every batch-1 call (the first launches of k_score_flag_count): duration of each nms kernel, the gap in front of it, and the
span from the first kernel's start to the last kernel's end -- what the device spends on nms when the launches are not waiting for
the host (hipGraph replay, or launches queued behind earlier work)."""
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute('select start, end, name from kernels order by start').fetchall()
NMS = ('k_score_flag_count', 'k_compact_decode', 'k_class_topk_iou', 'k_final_select')
calls, cur = [], None
for s, e, n in rows:
    k = next((x for x in NMS if x in n), None)
    if k == NMS[0]:
        cur = []
        calls.append(cur)
    if k and cur is not None:
        cur.append((k, s, e))
    elif cur is not None and not k:
        cur = None
calls = [x for x in calls if len(x) >= 3]
first = calls[3:50]                        # post_time.py: 3 warm-up + 50 timed batch-1 calls come first
med = lambda v: statistics.median(v) / 1e3
print(f'{len(first)} batch-1 nms calls, {len(first[0])} launches each (medians, us)')
for i, (k, _, _) in enumerate(first[0]):
    d = med([x[i][2] - x[i][1] for x in first])
    g = med([x[i][1] - x[i - 1][2] for x in first]) if i else 0.0
    print(f'  {k:22s} {d:7.2f}   gap in front {g:6.2f}')
print(f'  first start -> last end: {med([x[-1][2] - x[0][1] for x in first]):.2f}')
