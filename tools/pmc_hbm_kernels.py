#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of a traced command from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs,
--kernel-trace only): bytes per launch (gfx950 corrections as in tools/pmc_summary.py) and the rate against each kernel's own
duration from the same trace -> which passes are HBM-bound and how close to the ~8 TB/s peak they run.

  python tools/pmc_hbm_kernels.py <fetch.db> <write.db> <out.json> "<command description>"
"""
import json
import sqlite3
import sys


def clean(name):
    return name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def counters(db, counter):
    c = sqlite3.connect(db)
    out = {}
    for name, n, v in c.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name',
                                (counter,)).fetchall():
        a = out.setdefault(clean(name), [0, 0.0])
        a[0] += n
        a[1] += v
    return out


def durations(db):
    c = sqlite3.connect(db)
    out = {}
    for name, s, e in c.execute('select name, start, end from kernels').fetchall():
        a = out.setdefault(clean(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    return out


def main(fetch_db, write_db, out_path, desc):
    f, w, d = counters(fetch_db, 'FETCH_SIZE'), counters(write_db, 'WRITE_SIZE'), durations(fetch_db)
    rows = {}
    for k in f:
        n = f[k][0]
        fb, wb = f[k][1] * 1024 * 2 / n, w.get(k, [n, 0.0])[1] * 1024 / max(1, w.get(k, [n, 0.0])[0])
        us = d.get(k, [n, 0])[1] / max(1, d.get(k, [n, 0])[0]) / 1e3
        rows[k] = dict(launches=n, fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb), avg_us=round(us, 2),
                       tb_per_s=round((fb + wb) / max(us, 1e-9) / 1e6, 3), frac_hbm_peak=round((fb + wb) / max(us, 1e-9) / 1e6 / 8.0, 3))
    top = dict(sorted(rows.items(), key=lambda kv: -(kv[1]['fetch_bytes_per_launch'] + kv[1]['write_bytes_per_launch']) * kv[1]['launches'])[:24])
    out = {'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around `{desc}`',
           'units': 'rocprofv3 reports KB; bytes = KB*1024; gfx950: FETCH_SIZE x2 (128-B requests counted as 64 B), WRITE_SIZE 1:1; FETCH counts '
                    'L2->fabric requests incl. Infinity-Cache hits (upper bound on HBM reads); rate = bytes / the kernel\'s own average '
                    'duration in the FETCH pass; peak 8 TB/s',
           'kernels': top}
    json.dump(out, open(out_path, 'w'), indent=1)
    for k, v in list(top.items())[:14]:
        print(f'{k[:60]:60s} x{v["launches"]:5d} {v["fetch_bytes_per_launch"] / 1e6:8.2f} MB rd {v["write_bytes_per_launch"] / 1e6:8.2f} MB wr '
              f'{v["avg_us"]:8.1f} us {v["tb_per_s"]:6.2f} TB/s')


if __name__ == '__main__':
    main(*sys.argv[1:5])
