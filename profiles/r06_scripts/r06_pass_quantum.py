#!/usr/bin/env python3
"""Round quantisation of the persistent grids: time one dense hour of each stand-in net for pass sizes just below the cap.
    python profiles/r06_scripts/r06_pass_quantum.py"""
import os, sys, time
import numpy as np
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.getcwd())
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from inaspeechsegmenter_amd import _native as N, keras_model as KM, segmenter as S
from test_gpu_topologies import _mspec

ctx = N.Context(0)
ctx.set_precision_guard(False) if hasattr(ctx, 'set_precision_guard') else None
T = 360000 - 2
ctx.set_mspec(_mspec(np.random.default_rng(7), T))
rows = S._window_rows(T)
for net, nmel, ncls, seed, cands in (('vad', 21, 3, 1, [30360, 30352, 30344, 30336, 30328, 30320, 30304, 30288, 30272, 30240, 30200, 30100]),
                                     ('gender', 24, 2, 2, [25808, 25800, 25792, 25784, 25776, 25768, 25760, 25752, 25744, 25736, 25728, 25720])):
    layers, shp = KM.synthetic_ina_like(nmel, ncls, seed=seed)
    comp = KM.compile_layers(layers, shp)
    per = 4 * int(np.sum(comp.buf_elems))
    ctx.cnn_load(5, comp)
    ctx.cnn_probs(5, rows)
    for rep in range(2):
        for bc in cands:
            ctx.set_workspace_limit(per * bc + per // 2)
            ctx.cnn_probs(5, rows); ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); ctx.cnn_probs(5, rows); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
            print(f'{net} rep {rep} pass {bc}: {best * 1e3:.2f} ms ({-(-len(rows) // bc)} passes)', flush=True)
