#!/usr/bin/env python3
"""Inter-kernel gaps of a rocprofv3 --kernel-trace run (rocpd sqlite): for consecutive kernels on the device, gap = start(k+1) -
end(k).  Prints the distribution and the share of wall time spent between kernels.

  gap_summary.py results.db [top_n [marker]]
`marker` (a kernel-name substring that occurs once per step, e.g. k_sgd) restricts the analysis to the steady state: from the end
of its 3rd occurrence to the end of its last one, and reports per-step figures."""
import sqlite3
import sys

db = sys.argv[1]
top_n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
marker = sys.argv[3] if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
rows = c.execute('select start, end, name from kernels order by start').fetchall()
steps = None
if marker:
    marks = [e for s, e, n in rows if marker in n]
    if len(marks) >= 4:
        t0, t1 = marks[2], marks[-1]
        rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
        steps = len(marks) - 3
gaps, busy = [], 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 200_000 or marker:             # without a marker: ignore the pauses between bench phases
        gaps.append((g, n0.split('(')[0][-60:], n1.split('(')[0][-60:]))
    busy += e0 - s0
gs = sorted(g for g, _, _ in gaps)
# kernels of different streams may overlap (training: weight gradients on a side stream): device-busy time = UNION of the intervals
union, cur_end = 0, None
for s0, e0, _ in rows:
    if cur_end is None or s0 >= cur_end:
        union += e0 - s0
        cur_end = e0
    elif e0 > cur_end:
        union += e0 - cur_end
        cur_end = e0
busy += rows[-1][1] - rows[-1][0] if rows else 0
span_all = (rows[-1][1] - rows[0][0]) if rows else 0
tot_gap = span_all - union if marker else sum(g for g in gs if g > 0)
print(f'{len(rows)} kernels, kernel time {busy / 1e6:.2f} ms (sum over streams), device busy {union / 1e6:.2f} ms (union), '
      f'idle {tot_gap / 1e6:.2f} ms = {100 * tot_gap / max(1, tot_gap + union):.1f} % of the span')
if steps:
    span = rows[-1][1] - rows[0][0]
    print(f'  steady state: {steps} steps, {span / steps / 1e6:.2f} ms/step wall, {union / steps / 1e6:.2f} ms busy (union; '
          f'{busy / steps / 1e6:.2f} ms of kernel time, {busy / max(1, union):.2f} kernels in flight on average), '
          f'{tot_gap / steps / 1e6:.2f} ms idle, {len(rows) / steps:.0f} kernels/step')
    for lim in (5, 20, 100, 1000):
        part = sum(g for g in gs if 0 < g <= lim * 1000)
        print(f'  gaps <= {lim:4d} us: {part / steps / 1e6:.3f} ms/step ({sum(1 for g in gs if 0 < g <= lim * 1000) / steps:.0f} per step)')
if steps and top_n:
    from collections import Counter
    cnt, dur = Counter(), Counter()
    for s0, e0, n0 in rows:
        k = n0.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
        cnt[k] += 1
        dur[k] += e0 - s0
    print('  per step (steady state): launches, ms')
    for k, v in cnt.most_common(top_n):
        print(f'    {v / steps:7.1f} {dur[k] / steps / 1e6:8.3f}  {k}')
for q in (0.1, 0.5, 0.9, 0.99):
    print(f'  p{int(q * 100)} gap {gs[int(q * (len(gs) - 1))] / 1e3:.2f} us')
print(f'  overlapping pairs (negative gap): {sum(1 for g in gs if g < 0)}')
for g, a, b in sorted(gaps, reverse=True)[:top_n]:   # where the device waits for the host
    print(f'  {g / 1e3:8.1f} us  after {a[-44:]:44s} before {b[-44:]}')
