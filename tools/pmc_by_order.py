#!/usr/bin/env python3
"""Per-layer rocprofv3 --pmc counters of the ResNet-101 (vbx) program: the conv launches of the LAST pass of a
`tools/layer_prof.py --reps 1` run (one chunk), in launch order, joined with each layer's algorithmic bytes.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d D -o r -- python tools/layer_prof.py --minutes 2.1 --reps 1
    python tools/pmc_by_order.py D/.../r_results.db <windows> [pass from the end]"""
import os
import sqlite3
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from inaspeechsegmenter_amd import keras_model as KM, _native as N, vbx as V     # noqa: E402


def main():
    db, nwin = sys.argv[1], int(sys.argv[2])
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    order = next((k for k in ('dispatch_id', 'start', 'id') if k in cols), None)
    if order is None:
        raise SystemExit(f"no ordering column in counters_collection: {cols}")
    rows = list(c.execute(f"select {order}, kernel_name, counter_name, value, duration from counters_collection order by {order}"))
    disp = {}
    for o, k, cn, v, d in rows:
        e = disp.setdefault(o, {'k': k, 'us': d / 1e3})
        e[cn] = e.get(cn, 0.0) + v
    seq = [e for _, e in sorted(disp.items()) if 'conv' in e['k']]
    comp = KM.compile_resnet101(KM.synthetic_resnet101(0), V.FEAT_DIM, V.WINLEN, window_input=True)
    prog = np.asarray(comp.prog).reshape(-1, N.PROG_COLS)
    convs = [r for r in prog if r[N.C_OP] == N.OP_CONV]
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1        # which pass from the end (a remainder chunk may follow the full one)
    last = seq[len(seq) - back * len(convs):len(seq) - (back - 1) * len(convs)]
    cnames = sorted({k for e in last for k in e if k not in ('k', 'us')})
    print(f"# {len(seq)} conv dispatches, last {len(convs)} = one pass of {nwin} windows; counters {cnames} (KB where *_SIZE; FETCH_SIZE x2 = bytes on gfx950)\n")
    print("| # | kh kw s | Cin | Cout | Ho x Wo | res | kernel | us | alg read MB | alg write MB | " + " | ".join(cnames) + " |")
    print("|---|---|---|---|---|---|---|---|---|---|" + "---|" * len(cnames))
    for i, (r, e) in enumerate(zip(convs, last)):
        h, w, cin, ho, wo, cout, kh, kw, sh = [int(r[c]) for c in (N.C_H, N.C_W, N.C_CIN, N.C_HO, N.C_WO, N.C_COUT, N.C_KH, N.C_KW, N.C_SH)]
        res = r[N.C_RES] >= 0
        rd = 4.0 * (h * w * cin + (ho * wo * cout if res else 0)) * nwin / 1e6
        wr = 4.0 * ho * wo * cout * nwin / 1e6
        kn = e['k'].replace('void ', '').replace('(anonymous namespace)::', '').replace('issk::', '').split('(')[0][:34]
        vals = " | ".join(f"{e.get(cn, 0.0) / 1e3:.1f}" for cn in cnames)
        print(f"| {i} | {kh} {kw} {sh} | {cin} | {cout} | {ho}x{wo} | {int(res)} | `{kn}` | {e['us']:.1f} | {rd:.1f} | {wr:.1f} | {vals} |")


if __name__ == '__main__':
    main()
