#!/usr/bin/env python3
"""Per-layer kernel times of the ResNet-101 (vbx) program from a rocprofv3 --kernel-trace result (rocpd SQLite), joined
with each layer's algorithmic flops and activation bytes:  python tools/layer_times.py <results.db> [batch_windows]
Run on the GPU box right after:  rocprofv3 --kernel-trace -d D -o r -- python bench.py --workload vbx --steps 1 --warmup 0 --no-cpu-baseline"""
import os
import sqlite3
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from inaspeechsegmenter_amd import keras_model as KM, _native as N     # noqa: E402


def dispatches(db):
    c = sqlite3.connect(db)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    for v in ('kernels', 'rocpd_kernels'):
        if v in views:
            cols = [r[1] for r in c.execute(f"pragma table_info({v})")]
            if 'name' in cols and 'start' in cols and 'end' in cols:
                return list(c.execute(f"select name, start, end from {v} order by start"))
    raise SystemExit(f"no kernel dispatch view in {db}: {views}")


def main():
    db = sys.argv[1]
    bw = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    d = [(n, s, e) for n, s, e in dispatches(db) if any(k in n for k in ('conv_', 'pool_kernel', 'statpool'))]
    ends = [i for i, x in enumerate(d) if 'statpool' in x[0]]
    if len(ends) < 3:
        raise SystemExit(f"{len(ends)} batches found")
    mid = len(ends) // 2
    seq = d[ends[mid - 1] + 1:ends[mid] + 1]
    comp = KM.compile_resnet101(KM.synthetic_resnet101(0))
    prog = np.asarray(comp.prog).reshape(-1, N.PROG_COLS)
    rows = [r for r in prog if r[N.C_OP] in (N.OP_CONV,)]
    convs = [x for x in seq if 'conv_' in x[0]]
    convs = convs[1:] + convs[:1]                    # the embedding layer runs AFTER the statistics pooling: the slice starts with it
    print(f"# per-layer times, one batch of {bw} windows ({len(seq)} launches, {len(convs)} conv launches, {len(rows)} conv rows)\n")
    if len(convs) != len(rows):
        for n, s, e in seq:
            print(n[:60], (e - s) / 1e3)
        return
    print("| # | kh kw s | Cin | Cout | Ho x Wo | res | kernel | us | alg TFLOP/s | act GB/s | bound us (6 TB/s / 833 TF) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    agg = {}
    for i, (r, (n, s, e)) in enumerate(zip(rows, convs)):
        h, w, cin, ho, wo, cout, kh, kw, sh = [int(r[c]) for c in (N.C_H, N.C_W, N.C_CIN, N.C_HO, N.C_WO, N.C_COUT, N.C_KH, N.C_KW, N.C_SH)]
        res = r[N.C_RES] >= 0
        fl = 2.0 * ho * wo * cout * kh * kw * cin * bw
        by = 4.0 * (h * w * cin + ho * wo * cout * (2 if res else 1)) * bw
        us = (e - s) / 1e3
        kn = n.replace('void ', '').replace('(anonymous namespace)::', '').replace('issk::', '').split('(')[0][:40]
        bound = max(by / 6e12, fl / 833e12) * 1e6
        print(f"| {i} | {kh} {kw} {sh} | {cin} | {cout} | {ho}x{wo} | {int(res)} | `{kn}` | {us:.1f} | {fl / us / 1e6:.1f} | {by / us / 1e3:.0f} | {bound:.1f} |")
        k = (kh, kw, sh, cin, cout, ho, wo, int(res), kn)
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] += fl; a[3] += by; a[4] += bound
    print("\n## grouped by shape\n")
    print("| kh kw s Cin Cout HoxWo res | kernel | n | total us | alg TFLOP/s | act GB/s | bound us | x bound |")
    print("|---|---|---|---|---|---|---|---|")
    tot = tb = 0.0
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k[0]} {k[1]} {k[2]} {k[3]} {k[4]} {k[5]}x{k[6]} {k[7]} | `{k[8]}` | {a[0]} | {a[1]:.0f} | {a[2] / a[1] / 1e6:.1f} | {a[3] / a[1] / 1e3:.0f} | {a[4]:.0f} | {a[1] / a[4]:.2f} |")
        tot += a[1]; tb += a[4]
    print(f"\nconv total {tot:.0f} us per batch of {bw} windows; roofline bound (max of 6 TB/s activations, 833 TFLOP/s algorithmic) {tb:.0f} us")


if __name__ == '__main__':
    main()
