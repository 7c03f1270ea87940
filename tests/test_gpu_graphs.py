"""GPU parity of graph-shaped models (tests/graph_nets.py): functional Keras models that are not a chain -- residual adds,
inception-style concatenations (incl. along H / W), permutes, reshapes, several readers of one tensor, merges of feature vectors --
lowered onto chains (the usual kernels and fusions) joined by ISS_OP_ELT rows, against the Keras-semantics oracle: 1e-4 on
probabilities on the segmenter's overlapping windows and on scattered windows, in the split-operand and in the exact-f32 arithmetic.
`keras.models.load_model` (segmenter.py:129-131) accepts any of them; none may end in "no output"."""
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, segmenter as S, _native
import graph_nets as GN
from test_gpu_topologies import _mspec, _oracle_probs

pytestmark = pytest.mark.gpu


def _check(ctx, layers, shp, mspec, rows, tag, tol=1e-4):
    ctx.cnn_load(5, KM.compile_layers(layers, shp))
    probs, fin = ctx.cnn_probs(5, rows)
    ref, rfin = _oracle_probs(layers, mspec, shp[1], rows)
    assert np.array_equal(fin, rfin), tag
    err = np.abs(probs - ref).max()
    assert np.isfinite(probs).all() and err < tol, (tag, err)
    return probs, err


@pytest.mark.parametrize('name', sorted(GN.NETS))
def test_graph_model_parity(ctx, name):
    rng = np.random.default_rng(sum(map(ord, name)))
    T = 900
    mspec = _mspec(rng, T)
    mspec[400:402, 3] = -np.inf
    ctx.set_mspec(mspec)
    for nmel, ncls in ((21, 3), (24, 2)):
        layers, shp = GN.NETS[name](nmel, ncls, 5)
        rows = S._window_rows(T)
        probs, err = _check(ctx, layers, shp, mspec, rows, (name, nmel))
        scattered = np.sort(rng.choice(T - 68, 97, replace=False)).astype(np.int32)
        _check(ctx, layers, shp, mspec, scattered, (name, nmel, 'scattered'))
        # several passes instead of one: buffers are sized per pass, the merge rows work on whatever the pass holds
        prev = getattr(ctx, 'workspace_limit', None) or (24 << 30)
        per_slot = 4 * int(np.sum(KM.compile_layers(layers, shp).buf_elems))
        ctx.set_workspace_limit(max(per_slot * (len(rows) // 8 + 1), 64 << 20))       # (64 MiB is the floor: >= 2 passes for every net here)
        try:
            p2, _ = ctx.cnn_probs(5, rows)
        finally:
            ctx.set_workspace_limit(prev)
        assert np.abs(p2 - probs).max() < 4e-5, (name, nmel, np.abs(p2 - probs).max())
        for prec, tol in ((_native.PREC_F32, 2e-5), (_native.PREC_F16X3, 1e-4)):       # exact f32; fp16 halves (the product's default)
            ctx.set_precision(prec)
            try:
                _check(ctx, layers, shp, mspec, rows[:300], (name, nmel, prec), tol=tol)
            finally:
                ctx.set_precision(_native.PREC_BF16X3)
        print(f'{name} nmel {nmel}: max |dp| {err:.2e}')


# `ISS_GRAPH_FUZZ_N` / `ISS_GRAPH_FUZZ_BASE` widen the draw for a one-off soak run (profiles/r06_scripts/r06_graph_soak.sh)
@pytest.mark.parametrize('seed', range(int(os.environ.get('ISS_GRAPH_FUZZ_BASE', '0')),
                                       int(os.environ.get('ISS_GRAPH_FUZZ_BASE', '0')) + int(os.environ.get('ISS_GRAPH_FUZZ_N', '16'))))
def test_random_graph_parity(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    T = 600
    mspec = _mspec(rng, T)
    ctx.set_mspec(mspec)
    layers, shp = GN.random_graph(seed, 21 if seed % 2 else 24, 3 if seed % 2 else 2)
    rows = S._window_rows(T)
    _, err = _check(ctx, layers, shp, mspec, rows, ('random', seed))
    print(f'random graph {seed}: {sum(1 for L in layers if len(L["inputs"]) > 1)} merges, max |dp| {err:.2e}')


def test_merge_rows_are_refused_when_malformed(ctx):
    """iss_cnn_load checks ISS_OP_ELT rows (include/iss.h): a binary row without its second operand, a COPY outside the
    destination's channels, a PERMUTE that is not a permutation."""
    layers, shp = GN.NETS['residual'](21, 3, 5)
    comp = KM.compile_layers(layers, shp)
    k = next(i for i, R in enumerate(comp.prog) if R[_native.C_OP] == _native.OP_ELT)
    for col, val, msg in ((_native.C_RES, -1, 'second operand'), (_native.C_ACT, 9, 'unknown kind'), (_native.C_COUT, 16, 'keep the shape')):
        bad = KM.CompiledNet(comp.prog.copy(), comp.blob, comp.buf_elems, comp.in_shape, comp.out_dim, comp.flops_per_sample, True)
        bad.prog[k, col] = val
        with pytest.raises(_native.NativeError, match=msg):
            ctx.cnn_load(6, bad)
    layers, shp = GN.NETS['permute_reshape'](21, 3, 5)
    comp = KM.compile_layers(layers, shp)
    k = next(i for i, R in enumerate(comp.prog) if R[_native.C_OP] == _native.OP_ELT and R[_native.C_ACT] == _native.ELT_PERMUTE)
    bad = KM.CompiledNet(comp.prog.copy(), comp.blob, comp.buf_elems, comp.in_shape, comp.out_dim, comp.flops_per_sample, True)
    bad.prog[k, _native.C_KW] = bad.prog[k, _native.C_KH]
    with pytest.raises(_native.NativeError, match='permutation'):
        ctx.cnn_load(6, bad)


def test_precision_guard_probes_a_graph_model(ctx):
    """The library's first-call probe (iss_set_precision_guard) runs a graph-shaped network in its split mode and in exact f32 like any
    other program -- merge rows included -- and records the figure; fp16 halves (the product's default) pass on the residual stand-in."""
    rng = np.random.default_rng(77)
    T = 900
    mspec = _mspec(rng, T)
    ctx.set_mspec(mspec)
    layers, shp = GN.NETS['standin_residual'](21, 3, 5)
    rows = S._window_rows(T)
    ctx.set_precision(_native.PREC_F16X3)
    ctx.set_precision_guard(5e-4)
    try:
        ctx.cnn_load(5, KM.compile_layers(layers, shp))
        assert ctx.cnn_precision_info(5)['state'] == 'pending'
        probs, fin = ctx.cnn_probs(5, rows)
        info = ctx.cnn_precision_info(5)
        print('graph model:', info)
        assert info['state'] == 'passed' and info['mode'] == 'f16x3' and 0 <= info['max_dlogp'] < 5e-4 and info['slots'] > 100, info
        ref, rfin = _oracle_probs(layers, mspec, 21, rows)
        assert np.array_equal(fin, rfin) and np.abs(probs - ref).max() < 1e-4
    finally:
        ctx.set_precision_guard(0)
        ctx.set_precision(_native.PREC_BF16X3)
