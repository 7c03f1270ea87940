#!/usr/bin/env python3
"""Which kernel instantiations a topology of tests/topologies.py runs on, and how long each takes: one dense pass of both nets over
`--minutes` of log-mel rows with the HIP-event profile on (iss_prof_get_instance / iss_prof_get_row).
    python tools/topology_prof.py conv2_7x7 conv1_same [--minutes 20]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='+')
    ap.add_argument('--minutes', type=float, default=20.0)
    ap.add_argument('--diag', default='', help="kernel-selection switches for the run, e.g. 'no_fsame,no_ring'")
    args = ap.parse_args()
    from inaspeechsegmenter_amd import _native as N, keras_model as KM, segmenter as S
    import topologies as TP
    from test_gpu_topologies import _mspec
    ctx = N.Context(0)
    if args.diag:
        ctx.set_diag(args.diag)
        print(f'# diag = {args.diag}')
    T = int(args.minutes * 6000) - 2
    ctx.set_mspec(_mspec(np.random.default_rng(7), T))
    rows = S._window_rows(T)
    for name in args.names:
        for net, (layers, shp) in sorted(TP.nets(name).items()):
            comp = KM.compile_layers(layers, shp)
            ctx.cnn_load(5, comp)
            ctx.cnn_probs(5, rows)
            ctx.prof_enable(True)
            ctx.prof_reset()
            ctx.cnn_probs(5, rows)
            inst = ctx.prof_instances()
            tot = sum(k['ms'] for k in inst)
            oth_ms, oth_n, _ = ctx.prof_get(2)
            print(f"## {name} / {net}: {tot:.2f} ms in GEMM kernels, {comp.flops_per_sample * len(rows) / 1e9:.0f} GFLOP algorithmic; "
                  f"{oth_ms:.2f} ms in {oth_n} other launches (shared first layer, window statistics, pools, softmax)")
            prog = np.asarray(comp.prog).reshape(-1, N.PROG_COLS)
            for i, r in enumerate(prog):
                ms, nl = ctx.prof_get_row(i)
                if nl and r[N.C_OP] == N.OP_CONV:
                    print(f"  row {i}: conv {r[N.C_KH]}x{r[N.C_KW]} {r[N.C_CIN]}->{r[N.C_COUT]} {r[N.C_H]}x{r[N.C_W]}->{r[N.C_HO]}x{r[N.C_WO]}"
                          f"  {ms:7.3f} ms / {nl} launches")
                elif nl:
                    print(f"  row {i}: op {r[N.C_OP]}  {ms:7.3f} ms / {nl} launches")
            for k in sorted(inst, key=lambda k: -k['ms']):
                tf = k['flops'] / (k['ms'] * 1e-3) / 1e12 if k['ms'] > 0 else 0.0
                print(f"  {k['kernel']:<58s} {k['ms']:8.3f} ms {k['launches']:4d} launches {tf:7.1f} TF")
            ctx.prof_enable(False)


if __name__ == '__main__':
    main()
