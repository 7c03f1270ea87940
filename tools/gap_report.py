#!/usr/bin/env python3
"""Idle gaps on the GPU timeline of the LAST step of a bench run, from a rocprofv3 --kernel-trace result (rocpd SQLite):
python tools/gap_report.py <results.db> [first-kernel-substring [step-index]]   (default: sidekit_kernel starts a step; the last one is detailed)"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else 'sidekit_kernel'
    c = sqlite3.connect(db)
    d = list(c.execute("select name, start, end from kernels order by start"))
    starts = [i for i, x in enumerate(d) if first in x[0]]
    if not starts:
        raise SystemExit('no step start found')
    which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    for j, i0 in enumerate(starts):
        sq = d[i0:starts[j + 1]] if j + 1 < len(starts) else d[i0:]
        print(f"step {j}: {len(sq)} kernels, span {(max(x[2] for x in sq) - sq[0][1]) / 1e6:.2f} ms, sum of durations {sum(x[2] - x[1] for x in sq) / 1e6:.2f} ms")
    i0 = starts[which]
    seq = d[i0:starts[which + 1]] if which != -1 and which + 1 < len(starts) else d[i0:]
    t0, t1 = seq[0][1], max(x[2] for x in seq)
    busy, cur_end, gaps = 0, seq[0][1], []
    for n, s, e in seq:
        if s > cur_end:
            gaps.append((s - cur_end, n))
        busy += max(0, e - max(s, cur_end))
        cur_end = max(cur_end, e)
    print(f"last step: {len(seq)} kernels, span {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    print("largest gaps (ms, kernel that follows):")
    for g, n in sorted(gaps, reverse=True)[:12]:
        print(f"  {g / 1e6:8.3f}  {n[:70]}")
    hist = {}
    for g, n in gaps:
        k = n.split('(')[0][-50:]
        a = hist.setdefault(k, [0, 0])
        a[0] += 1; a[1] += g
    print("idle time by following kernel:")
    for k, (cnt, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:10]:
        print(f"  {tot / 1e6:8.3f} ms in {cnt:4d} gaps before {k}")


if __name__ == '__main__':
    main()
