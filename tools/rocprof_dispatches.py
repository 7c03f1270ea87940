#!/usr/bin/env python3
"""List kernel dispatches (name, grid, duration) of a rocprofv3 rocpd database, aggregated by (kernel, grid)."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
filt = sys.argv[2] if len(sys.argv) > 2 else 'conv'
rows = c.execute("select name, grid_x, grid_y, count(*), avg(duration), sum(duration) from kernels group by name, grid_x, grid_y order by sum(duration) desc").fetchall()
tot = sum(r[5] for r in rows if filt in r[0])
for name, gx, gy, n, avg, s in rows:
    if filt not in name: continue
    nm = re.sub(r'\(anonymous namespace\)::', '', name); nm = re.sub(r'^void ', '', nm); nm = nm.split('(')[0]
    print(f"{nm:42s} grid=({gx//256:6d},{gy:3d}) n={n:4d} avg={avg/1e3:9.1f}us  {100*s/tot:5.1f}%")
