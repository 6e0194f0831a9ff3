#!/usr/bin/env python3
"""Per-kernel instruction mix from a rocprofv3 counter_collection.csv (tools/pmc_valu.sh)."""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')
    k = k[:k.index('(')] if '(' in k else k
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    seen[k].add(r['Dispatch_Id'])
rows = []
for k, c in agg.items():
    mf = c.get('SQ_INSTS_MFMA', 0.0)
    va = c.get('SQ_INSTS_VALU', 0.0)
    rows.append((c.get('SQ_BUSY_CU_CYCLES', 0.0), k, len(seen[k]), va, mf, c.get('SQ_INSTS_SALU', 0.0), c.get('SQ_INSTS_LDS', 0.0),
                 c.get('SQ_INSTS_VMEM_RD', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows) or 1.0
with open(dst, 'w') as f:
    f.write('share of busy CU cycles | launches | VALU (incl. MFMA) | MFMA | other VALU per MFMA | SALU | LDS | VMEM_RD | kernel\n')
    for busy, k, n, va, mf, sa, ld, vm, mb in rows[:40]:
        other = (va - mf) / mf if mf else float('nan')
        f.write(f'{busy / tot:6.1%} {n:6d} {va:14.0f} {mf:12.0f} {other:8.2f} {sa:14.0f} {ld:12.0f} {vm:12.0f}  {k[:110]}\n')
print(open(dst).read())
