#!/usr/bin/env python3
"""Inter-kernel gaps of a rocprofv3 --kernel-trace run (rocpd sqlite): for consecutive kernels on the device, gap = start(k+1) -
end(k).  Prints the distribution and the share of wall time spent between kernels."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute('select start, end, name from kernels order by start').fetchall()
gaps, busy = [], 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 200_000:                       # ignore the pauses between bench phases
        gaps.append((g, n0.split('(')[0][-60:], n1.split('(')[0][-60:]))
    busy += e0 - s0
gs = sorted(g for g, _, _ in gaps)
tot_gap = sum(gs)
print(f'{len(rows)} kernels, busy {busy / 1e6:.2f} ms, gaps(<200us) {tot_gap / 1e6:.2f} ms = {100 * tot_gap / (tot_gap + busy):.1f} % of busy+gap')
for q in (0.1, 0.5, 0.9, 0.99):
    print(f'  p{int(q * 100)} gap {gs[int(q * (len(gs) - 1))] / 1e3:.2f} us')
neg = sum(1 for g in gs if g < 0)
print(f'  overlapping pairs (negative gap): {neg}')
