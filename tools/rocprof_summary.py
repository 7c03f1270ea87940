#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.x, rocpd SQLite output) result into the short per-kernel summary
that is committed under profiles/:  python tools/rocprof_summary.py <results.db> [out.md] [title]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    name = m.group(1) if m else name
    return name if len(name) <= 90 else name[:87] + '...'


def main():
    db = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    title = sys.argv[3] if len(sys.argv) > 3 else db
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = [f"# rocprofv3 --kernel-trace --stats: {title}", "",
             "| kernel | calls | total ms | avg us | % of GPU kernel time |", "|---|---|---|---|---|"]
    for name, calls, tot, avg, pct in rows[:14]:
        lines.append(f"| `{short(name)}` | {calls} | {tot / 1e3:.3f} | {avg:.2f} | {pct:.2f} |")
    rest = rows[14:]
    if rest:
        lines.append(f"| ({len(rest)} more kernels, mostly torch elementwise ops of the synthetic-audio generator) | "
                     f"{sum(r[1] for r in rest)} | {sum(r[2] for r in rest) / 1e3:.3f} | | {sum(r[4] for r in rest):.2f} |")
    conv = [r for r in rows if 'conv_igemm' in r[0]]
    if conv:
        calls = sum(r[1] for r in conv)
        tot = sum(r[2] for r in conv)
        lines += ["", f"conv_igemm_kernel, all instantiations: {calls} launches, {tot / 1e3:.3f} ms total, "
                      f"average {tot / calls / 1e3:.4f} ms per launch"]
    txt = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
