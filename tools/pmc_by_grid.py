#!/usr/bin/env python3
"""Per (kernel, grid) averages of rocprofv3 --pmc counters: python tools/pmc_by_grid.py <results.db> [name-filter]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else 'conv_'
    q = ("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, grid_size, counter_name")
    rows = {}
    for k, g, cn, n, v, d in c.execute(q):
        if filt not in k:
            continue
        k = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', k)).split('(')[0][:60]
        r = rows.setdefault((k, g), {'n': n, 'us': d / 1e3})
        r[cn] = v
    tot = sum(r['n'] * r['us'] for r in rows.values())
    for (k, g), r in sorted(rows.items(), key=lambda kv: -kv[1]['n'] * kv[1]['us']):
        cs = ' '.join(f"{cn}={v:.4g}" for cn, v in r.items() if cn not in ('n', 'us'))
        print(f"{k:58s} grid={int(g):9d} n={r['n']:5d} avg={r['us']:8.1f}us {100 * r['n'] * r['us'] / tot:5.1f}%  {cs}")


if __name__ == '__main__':
    main()
