#!/usr/bin/env python3
"""The random DAG nets of a soak that pass 1e-4 by a hair: the same nets in the three arithmetic modes (is it the arithmetic or a bug?).
    python profiles/r06_scripts/r06_graph_outliers.py 9088 9093"""
import os, sys
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from inaspeechsegmenter_amd import _native as N, keras_model as KM, segmenter as S
import graph_nets as GN
from test_gpu_topologies import _mspec, _oracle_probs

ctx = N.Context(0)
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(1000 + seed)
    T = 600
    mspec = _mspec(rng, T)
    ctx.set_mspec(mspec)
    layers, shp = GN.random_graph(seed, 21 if seed % 2 else 24, 3 if seed % 2 else 2)
    rows = S._window_rows(T)
    comp = KM.compile_layers(layers, shp)
    ref, rfin = _oracle_probs(layers, mspec, shp[1], rows)
    from oracle import keras_cnn as ocnn
    out = {}
    for guard in (0, 5e-4):                                              # 0 = off (the shared test context), 5e-4 = the library default
        ctx.set_precision_guard(guard)
        for name, prec in (('bf16x3', N.PREC_BF16X3), ('f16x3', N.PREC_F16X3), ('f32', N.PREC_F32)):
            ctx.set_precision(prec)
            ctx.cnn_load(5, comp)
            p, f = ctx.cnn_probs(5, rows)
            info = ctx.cnn_precision_info(5) if hasattr(ctx, 'cnn_precision_info') else {}
            out[f'{name}{"+guard" if guard else ""}'] = np.abs(p - ref).max()
            if guard: out[f'{name}+guard'] = f"{np.abs(p - ref).max():.2e} [{info.get('mode', '?')}, probe {(info.get('max_dlogp') or float('nan')):.1e}]"
    ctx.set_precision(N.PREC_BF16X3)
    print(f'random graph {seed}: max |dp| vs oracle ' + ', '.join(f'{k} {v if isinstance(v, str) else format(v, ".2e")}' for k, v in out.items()) +
          f'; merges {sum(1 for L in layers if len(L["inputs"]) > 1)}, min p in the oracle {ref[rfin].min():.1e}')
