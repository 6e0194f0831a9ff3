#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of a bench.py command into the
HBM-side bytes per conv launch that bench.py reports as `roofline.traffic`.

  python tools/pmc_summary.py <fetch.db> <write.db> <out.json> "<command description>"

Units / corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): rocprofv3 reports KB (bytes = KB*1024); on gfx950 FETCH_SIZE
counts 128-B requests as 64 B -> x2 for wide coalesced reads; WRITE_SIZE is 1:1.  FETCH counts L2->fabric requests (Infinity-Cache
hits included), so it is an upper bound on HBM reads.
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute('select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name',
                     (counter,)).fetchall()
    out = {}
    for name, n, v in rows:
        name = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        a = out.setdefault(name, [0, 0.0])
        a[0] += n
        a[1] += v
    return out


def main(fetch_db, write_db, out_path, desc):
    f, w = per_kernel(fetch_db, 'FETCH_SIZE'), per_kernel(write_db, 'WRITE_SIZE')
    conv = [k for k in f if k.startswith('conv_igemm_f32') or k.startswith('conv_igemm_pers') or k.startswith('conv_wdma_f32') or k.startswith('conv_wave_f32') or k.startswith('conv1x1_ws') or k.startswith('k_stem_pool') or k.startswith('conv_splitk_reduce')]
    main_k = [k for k in conv if not k.startswith('conv_splitk_reduce')]
    launches = sum(f[k][0] for k in main_k)
    fetch_b = sum(f[k][1] for k in conv) * 1024 * 2
    write_b = sum(w[k][1] for k in conv if k in w) * 1024
    per = {k: {'launches': f[k][0], 'FETCH_SIZE': f[k][1], 'WRITE_SIZE': w.get(k, [0, 0.0])[1]}
           for k in sorted(f, key=lambda k: -f[k][1])[:16]}
    out = {'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) around `{desc}`',
           'units': 'rocprofv3 reports KB; bytes = KB*1024; gfx950: FETCH_SIZE x2 (128-B requests counted as 64 B), WRITE_SIZE 1:1; '
                    'FETCH counts L2->fabric requests incl. Infinity-Cache hits (upper bound on HBM reads)',
           'conv_kernels': {'launches': launches, 'fetch_bytes_corrected_per_launch': fetch_b / launches,
                            'write_bytes_per_launch': write_b / launches, 'traffic_bytes_per_launch': (fetch_b + write_b) / launches},
           'per_kernel_KB_raw': per}
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps(out['conv_kernels']))


if __name__ == '__main__':
    main(*sys.argv[1:5])
