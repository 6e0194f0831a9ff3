#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel stats table committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for r in rows:
        name = r[0].replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0]
        lines.append(f'| {name} | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / tot:.1f} |')
    text = '\n'.join(lines)
    if out:
        open(out, 'w').write(text + '\n')
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:])
