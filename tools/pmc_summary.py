#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd SQLite result.

    python tools/pmc_summary.py <results.db> [name-filter ...]  ->  JSON on stdout:
    {kernel_short_name: {"launches": n, "avg_duration_us": d, counter: avg_per_launch, ...}}
Run on the GPU box right after the rocprofv3 pass (the .db files are too big to bring back)."""
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:80]


def main():
    db = sys.argv[1]
    filt = sys.argv[2:] or ['conv_', 'conv1_', 'sidekit_kernel', 'pool_kernel', 'patch_stats', 'softmax_kernel', 'vbx_', 'statpool']
    c = sqlite3.connect(db)
    out = {}
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration), avg(grid_size), avg(vgpr_count), "
         "avg(accum_vgpr_count), avg(lds_block_size) from counters_collection group by kernel_name, counter_name")
    for kname, cname, n, v, dur, grid, vg, ag, lds in c.execute(q):
        if not any(f in kname for f in filt):
            continue
        d = out.setdefault(short(kname), {})
        d['launches'] = n
        d['avg_duration_us'] = dur / 1e3
        d['avg_grid_threads'] = grid
        d['vgpr'] = vg
        d['agpr'] = ag
        d['lds_bytes'] = lds
        d[cname] = v
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
