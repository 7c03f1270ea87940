#!/usr/bin/env python3
"""Per-layer times of the ResNet-101 (vbx) program from the library's own HIP-event brackets (iss_prof_get_row), joined with
each layer's algorithmic flops and activation bytes.  Unlike tools/layer_times.py this needs no rocprofv3 (whose kernel
trace slows every launch of this program by ~2x on some boxes) and measures warm launches:
    python tools/layer_prof.py [--minutes 20] [--reps 3] [--workspace-mb N]        (on the GPU box)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from inaspeechsegmenter_amd import keras_model as KM, _native as N, vbx as V     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=20.0)
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--workspace-mb', type=int, default=0)
    args = ap.parse_args()
    ctx = N.Context(0)
    if args.workspace_mb:
        ctx.set_workspace_limit(args.workspace_mb << 20)
    rng = np.random.default_rng(0)
    n = int(args.minutes * 60 * 16000)
    pcm = (rng.normal(0, 3000, n)).astype(np.int16)
    fe = V.FeatureExtractor(ctx)
    ex = V.VBxExtractor(ctx, KM.synthetic_resnet101(0))
    fea = fe(pcm, to_host=False)
    starts = list(range(0, len(fea) - V.WINLEN, V.STEP))
    nid = ex._net_for(V.WINLEN, window=True)
    ctx.vbx_embed(nid, starts)                       # warm-up
    ctx.prof_enable(True)
    ctx.prof_reset()
    for _ in range(args.reps):
        ctx.vbx_embed(nid, starts)
    comp = KM.compile_resnet101(KM.synthetic_resnet101(0), V.FEAT_DIM, V.WINLEN, window_input=True)
    prog = np.asarray(comp.prog).reshape(-1, N.PROG_COLS)
    bw = len(starts) * args.reps
    print(f"# per-layer HIP-event times, {len(starts)} windows x {args.reps} passes ({args.minutes:g} min of audio), us per 512 windows\n")
    print("| row | kh kw s | Cin | Cout | Ho x Wo | res | launches | us / 512 win | alg TFLOP/s | act GB/s | bound us (6 TB/s / 833 TF) | x bound |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    agg = {}
    tot = tb = 0.0
    rows = []                                        # one entry per conv row: shape, time, algorithmic flops / activation bytes
    for i, r in enumerate(prog):
        if r[N.C_OP] != N.OP_CONV:
            continue
        ms, nl = ctx.prof_get_row(i)
        h, w, cin, ho, wo, cout, kh, kw, sh = [int(r[c]) for c in (N.C_H, N.C_W, N.C_CIN, N.C_HO, N.C_WO, N.C_COUT, N.C_KH, N.C_KW, N.C_SH)]
        res = r[N.C_RES] >= 0
        rows.append(dict(i=i, kh=kh, kw=kw, sh=sh, cin=cin, cout=cout, ho=ho, wo=wo, res=int(res), nl=nl, us=ms * 1e3 * 512 / bw,
                         fl=2.0 * ho * wo * cout * kh * kw * cin * 512, rd=4.0 * h * w * cin * 512, rr=4.0 * ho * wo * cout * 512 * int(res),
                         wr=4.0 * ho * wo * cout * 512, dual=r[N.C_DUALW] > 0, note=''))
    for k, e in enumerate(rows):                     # launches that compute two rows: account them together
        if e['nl'] == 0 and k + 1 < len(rows) and rows[k + 1]['dual'] and rows[k + 1]['nl']:
            t = rows[k + 1]                          # projection shortcut inside the next row's two-source GEMM: its output never exists
            t['fl'] += e['fl']; t['rd'] += 4.0 * e['ho'] * e['wo'] * e['cin'] * 512; t['rr'] = 0.0
            e['note'] = f"(one GEMM with row {t['i']})"; t['note'] = 'two-source'
        elif e['nl'] == 0 and k > 0 and rows[k - 1]['nl'] and e['kh'] == 1 and e['cin'] == rows[k - 1]['cout']:
            t = rows[k - 1]                          # reduction chained behind the previous row's expansion: x' is not read back
            t['fl'] += e['fl']; t['wr'] += e['wr']
            e['note'] = f"(chained launch with row {t['i']})"; t['note'] = 'chained'
    for e in rows:
        if e['nl'] == 0:
            print(f"| {e['i']} | {e['kh']} {e['kw']} {e['sh']} | {e['cin']} | {e['cout']} | {e['ho']}x{e['wo']} | {e['res']} | 0 | {e['note']} | | | | |")
            continue
        by = e['rd'] + e['rr'] + e['wr']
        bound = max(by / 6e12, e['fl'] / 833e12) * 1e6
        us = e['us']
        print(f"| {e['i']} | {e['kh']} {e['kw']} {e['sh']} | {e['cin']} | {e['cout']} | {e['ho']}x{e['wo']} | {e['res']} | {e['nl']} | {us:.1f} {e['note']} | "
              f"{e['fl'] / max(us, 1e-9) / 1e6:.1f} | {by / max(us, 1e-9) / 1e3:.0f} | {bound:.1f} | {us / bound:.2f} |")
        k = (e['kh'], e['kw'], e['sh'], e['cin'], e['cout'], e['ho'], e['wo'], e['res'], e['note'])
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] += e['fl']; a[3] += by; a[4] += bound
    print("\n## grouped by shape\n")
    print("| kh kw s Cin Cout HoxWo res | n | total us | alg TFLOP/s | act GB/s | bound us | x bound |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| {k[0]} {k[1]} {k[2]} {k[3]} {k[4]} {k[5]}x{k[6]} {k[7]} {k[8]} | {a[0]} | {a[1]:.0f} | {a[2] / a[1] / 1e6:.1f} | {a[3] / a[1] / 1e3:.0f} | {a[4]:.0f} | {a[1] / a[4]:.2f} |")
        tot += a[1]; tb += a[4]
    print(f"\nconv total {tot:.0f} us per 512 windows; roofline bound (max of 6 TB/s activations, 833 TFLOP/s algorithmic) {tb:.0f} us")


if __name__ == '__main__':
    main()
