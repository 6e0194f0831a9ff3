#!/usr/bin/env python3
"""Which kernels run on OTHER queues while a training forward is on the device?  (rocprofv3 --kernel-trace db; forward window = from
`k_nchw_to_nhwc4` to the first `k_match` after it.)"""
import sqlite3, sys
from collections import Counter
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info('kernels')").fetchall()]
print('columns:', cols)
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = c.execute(f'select start, end, name, {qcol or 0} from kernels order by start').fetchall()
starts = [r for r in rows if 'k_nchw_to_nhwc4' in r[2]]
match = [r for r in rows if 'k_match' in r[2]]
for w, s in enumerate(starts):
    m = next((x for x in match if x[0] > s[0]), None)
    if m is None:
        break
    t0, t1 = s[0], m[0]
    inside = [r for r in rows if r[1] > t0 and r[0] < t1]
    per_q = Counter()
    names = Counter()
    for r in inside:
        per_q[r[3]] += min(r[1], t1) - max(r[0], t0)
        if r[3] != s[3]:
            names[r[2].replace('(anonymous namespace)::', '').split('(')[0][:50]] += min(r[1], t1) - max(r[0], t0)
    print(f'forward {w}: {(t1 - t0) / 1e6:.2f} ms; busy per queue (ms): ' + ', '.join(f'{q}: {v / 1e6:.2f}' for q, v in per_q.most_common()) +
          ' | other queues: ' + ', '.join(f'{n} {v / 1e6:.2f}' for n, v in names.most_common(4)))
