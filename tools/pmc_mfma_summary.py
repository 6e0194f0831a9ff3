#!/usr/bin/env python3
"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (counters SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY, --kernel-trace only):
    python tools/pmc_mfma_summary.py <counter_collection.csv> <out.json> "<command description>"
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); GRBM_GUI_ACTIVE is summed over the 8 XCDs
(checked against the kernel's wall time: 1.39e7 / 8 cycles over 792 us = 2.19 GHz)."""
import collections
import csv
import json
import sys


def main(path, out, desc):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            calls[k] += 1
    rows = {}
    for k, v in agg.items():
        g = v.get('GRBM_GUI_ACTIVE', 0.0)
        if g <= 0:
            continue
        simd_cycles = g / 8.0 * 1024.0
        wc = v.get('SQ_WAVE_CYCLES', 0.0)
        rows[k] = dict(launches=calls[k], gui_active_cycles_per_launch=round(g / 8.0 / max(calls[k], 1)),
                       mfma_busy_frac=round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / simd_cycles, 4),
                       mfma_insts_per_launch=round(v.get('SQ_INSTS_MFMA', 0.0) / max(calls[k], 1)),
                       waves_resident_per_simd=round(wc * 4.0 / simd_cycles, 3),
                       wave_time_split=dict(wait_any=round(v.get('SQ_WAIT_ANY', 0.0) / wc, 3) if wc else None,
                                            wait_inst_any=round(v.get('SQ_WAIT_INST_ANY', 0.0) / wc, 3) if wc else None,
                                            active_inst_any=round(v.get('SQ_ACTIVE_INST_ANY', 0.0) / wc, 3) if wc else None))
    top = dict(sorted(rows.items(), key=lambda kv: -kv[1]['gui_active_cycles_per_launch'] * kv[1]['launches'])[:12])
    json.dump(dict(source=desc, note=__doc__.split('busy fraction')[1].strip() if 'busy fraction' in __doc__ else '', kernels=top),
              open(out, 'w'), indent=1)
    for k, r in top.items():
        print(f"{k[:70]:70s} x{r['launches']:5d} mfma busy {r['mfma_busy_frac']:.3f} waves/simd {r['waves_resident_per_simd']}")


if __name__ == '__main__':
    main(*sys.argv[1:4])
