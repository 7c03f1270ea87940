"""GPU: randomised small-CNN topologies against the Keras-semantics oracle.

The reference's three nets are un-vendored release assets whose `model_config` is unknown (remote_utils.py:4-15; loaded at
segmenter.py:129-131, run at :163): the engine must be right for whatever comes out of that file, not for the shapes the
kernels were tuned on.  tests/topologies.py moves one design choice at a time; this file draws WHOLE nets from a grammar
(filter shapes 1x3 ... 7x7, 'valid' / 'same', strides, 16 ... 128 channels incl. 48 / 96, BatchNorm before / after the
activation, elu / leaky relu / selu / softplus, max / average / overlapping pools, flatten or global pooling heads) with a fixed seed and holds every one of them
to 1e-4 on probabilities, on the segmenter's overlapping window list (shared first layer where it applies), on scattered
windows, and in the exact-f32 mode.

`ISS_FUZZ_NNETS` / `ISS_FUZZ_BASE` / `ISS_FUZZ_T` (frames of the recording) widen the draw for a one-off soak run (profiles/r05_scripts/r05_fuzz_soak.sh); the default
48 nets from base 9000 are what the suite runs."""
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, segmenter as S, _native
import topologies as TP
from test_gpu_topologies import _mspec, _oracle_probs

pytestmark = pytest.mark.gpu
NNETS = int(os.environ.get('ISS_FUZZ_NNETS', '48'))
BASE = int(os.environ.get('ISS_FUZZ_BASE', '9000'))
FRAMES = int(os.environ.get('ISS_FUZZ_T', '700'))


def random_spec(rng):
    """One net as a tests/topologies.py spec: 2-4 conv blocks, shrinking with pools / strides while the map stays >= 3 x 2."""
    spec = []
    h, w = 68, None                                   # (w depends on nmel: the grammar only uses choices valid for 21 and 24 columns)
    wmin = 21
    ch = int(rng.choice([16, 32, 48, 64, 64, 64]))
    nblocks = int(rng.integers(2, 5))
    for b in range(nblocks):
        shapes = [(3, 3), (5, 3), (4, 5), (3, 5), (5, 5), (7, 7), (7, 7), (5, 5), (4, 5), (1, 3), (3, 1), (2, 2)] if b else [(4, 5), (3, 3), (5, 3), (5, 5), (3, 5)]
        shapes = [(kh, kw) for kh, kw in shapes if kh <= h - 2 and kw <= wmin - 1]
        kh, kw = shapes[int(rng.integers(0, len(shapes)))]
        pad = 'same' if rng.random() < 0.4 else 'valid'
        stride = 2 if (b and rng.random() < 0.15 and h >= 12 and wmin >= 8) else 1
        spec.append(('conv', kh, kw, ch, pad, stride))
        h = -(-h // stride) if pad == 'same' else (h - kh) // stride + 1
        wmin = -(-wmin // stride) if pad == 'same' else (wmin - kw) // stride + 1
        spec.append([('bn_relu',), ('relu_bn',), ('relu',), ('bn_relu',), ('bn_relu',), ('elu', 0.8), ('leaky_relu', 0.1), ('selu',),
                     ('softplus',)][int(rng.integers(0, 9))])
        r = rng.random()
        if r < 0.5 and h >= 6 and wmin >= 4:
            kind = 'maxpool' if rng.random() < 0.75 else 'avgpool'
            ph, pw = [(2, 2), (2, 1), (1, 2), (2, 2)][int(rng.integers(0, 4))]
            if rng.random() < 0.15 and h >= 8 and wmin >= 6:
                spec.append((kind, 3, 3, 2, 2))       # overlapping
                h, wmin = (h - 3) // 2 + 1, (wmin - 3) // 2 + 1
            else:
                spec.append((kind, ph, pw))
                h, wmin = h // ph, wmin // pw
        if h < 4 or wmin < 3:
            break
        ch = int(min(128, ch * int(rng.choice([1, 1, 2])))) if rng.random() < 0.8 else int(rng.choice([32, 64, 96, 128]))
    head = int(rng.integers(0, 3))
    if head == 0:
        spec += [('flatten',), ('dense', int(rng.choice([64, 128, 192]))), ('drop',)]
    elif head == 1:
        spec += [('gap',), ('dense', 64)]
    else:
        spec += [('flatten',), ('dense', 192, 'linear'), ('bn_relu',), ('dense', 96, 'tanh')]
    return spec


@pytest.mark.parametrize('k', range(NNETS))
def test_random_topology(ctx, k):
    rng = np.random.default_rng(BASE + k)
    spec = random_spec(rng)
    nmel, ncls = (21, 3) if k % 2 == 0 else (24, 2)
    layers, shp = TP.build(spec, nmel, ncls, seed=100 + k)
    T = FRAMES
    mspec = _mspec(rng, T)
    mspec[300:302, 4] = -np.inf
    ctx.set_mspec(mspec)
    ctx.cnn_load(5, KM.compile_layers(layers, shp))
    rows = S._window_rows(T)
    ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
    ctx.prof_enable(True)
    ctx.prof_reset()
    probs, fin = ctx.cnn_probs(5, rows)
    insts = [i['kernel'] for i in ctx.prof_instances()]
    ctx.prof_enable(False)
    err = np.abs(probs - ref).max()
    scat = np.sort(rng.integers(0, T - 68, 48)).astype(np.int32)
    p2, f2 = ctx.cnn_probs(5, scat)
    r2, rf2 = _oracle_probs(layers, mspec, nmel, scat)
    err2 = np.abs(p2 - r2).max()
    ctx.set_precision(_native.PREC_F32)
    try:
        p3, f3 = ctx.cnn_probs(5, rows)
    finally:
        ctx.set_precision(_native.PREC_BF16X3)
    err3 = np.abs(p3 - ref).max()
    ctx.set_precision(_native.PREC_F16X3)                                # the library default: fp16 halves / exact f32 / bf16 halves per layer
    try:
        p4, f4 = ctx.cnn_probs(5, rows)
        p5, f5 = ctx.cnn_probs(5, scat)
    finally:
        ctx.set_precision(_native.PREC_BF16X3)
    err4, err5 = np.abs(p4 - ref).max(), np.abs(p5 - r2).max()
    print(f'net {k}: {spec}\n   kernels {sorted(set(insts))}\n   overlapping {err:.2e}, scattered {err2:.2e}, exact-f32 {err3:.2e}, fp16 halves {err4:.2e} / {err5:.2e}')
    assert np.array_equal(fin, rfin) and np.array_equal(f2, rf2) and np.array_equal(f3, rfin) and np.array_equal(f4, rfin) and np.array_equal(f5, rf2)
    assert err < 1e-4 and err2 < 1e-4 and err3 < 1e-4 and err4 < 1e-4 and err5 < 1e-4, (k, spec, err, err2, err3, err4, err5)
