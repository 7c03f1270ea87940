"""GPU parity: CNN engine (patch gather + z-norm + implicit-GEMM conv/dense on MFMA + fused / stand-alone
pooling + softmax) vs the Keras-semantics oracle, on seeded synthetic weights, in both arithmetic
modes: split-bf16 MFMA (default; operand error 2^-16) and exact-f32 MFMA.  Tolerance 1e-3 on
probabilities is the north-star bound; the tests hold both modes to 1e-4 (observed ~1e-6 .. 1e-5)."""
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, segmenter as S, _native
from oracle import keras_cnn as ocnn, segment as oseg
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
PROB_TOL = 1e-3


@pytest.fixture(params=['bf16x3', 'f16x3', 'f32'])
def prec(request, ctx):
    ctx.set_precision({'f32': _native.PREC_F32, 'f16x3': _native.PREC_F16X3}.get(request.param, _native.PREC_BF16X3))
    yield request.param
    ctx.set_precision(_native.PREC_BF16X3)


def _rand_conv(rng, kh, kw, cin, cout, strides=(1, 1), padding='valid', act='linear', bias=True):
    return dict(type='conv2d', W=rng.normal(0, np.sqrt(2.0 / (kh * kw * cin)), (kh, kw, cin, cout)).astype(np.float32),
                b=rng.normal(0, 0.1, cout).astype(np.float32) if bias else None, strides=strides, padding=padding,
                activation=act)


def _rand_bn(rng, c):
    return dict(type='batchnorm', gamma=rng.uniform(0.5, 1.5, c).astype(np.float32), beta=rng.normal(0, 0.2, c).astype(np.float32),
                mean=rng.normal(0, 0.2, c).astype(np.float32), var=rng.uniform(0.5, 1.5, c).astype(np.float32), eps=1e-3)


def _rand_dense(rng, i, o, act='linear'):
    return dict(type='dense', W=rng.normal(0, np.sqrt(1.0 / i), (i, o)).astype(np.float32),
                b=rng.normal(0, 0.1, o).astype(np.float32), activation=act)


# the fp16-operand instantiations of the segmenter nets' kernels, as iss_prof_get_instance spells them (template arguments)
_F16_KERNELS = {'conv_x3_wq_kernel<5,3,true,true>', 'conv_x3_wq3h_kernel<0,true,true>', 'conv_x3_wq3h_kernel<1,true,true>',
                'conv_dhl_kernel<true,8>'}


def _mspec(rng, T):
    return (rng.normal(-3, 2, (T, 24))).astype(np.float32)


def _oracle_probs(layers, mspec, nmel, rows):
    patches = np.stack([mspec[r:r + 68, :nmel] for r in rows])
    flat = patches.reshape(len(rows), -1)
    with np.errstate(invalid='ignore', divide='ignore'):
        z = (flat - flat.mean(axis=1, keepdims=True)) / flat.std(axis=1, keepdims=True)
    fin = np.all(np.isfinite(z), axis=1)
    z = np.where(fin[:, None], z, 0).reshape(len(rows), 68, nmel, 1).astype(np.float32)
    p = ocnn.forward(layers, z)
    p[~fin] = 0.5
    return p, fin


TOPOLOGIES = {
    'conv_valid_relu_dense': lambda r, h: [_rand_conv(r, 4, 5, 1, 16, act='relu'), dict(type='flatten'),
                                           _rand_dense(r, 65 * (h - 4) * 16, 3, 'softmax')],
    'same_pad_stride2_bn_before_act': lambda r, h: [_rand_conv(r, 3, 3, 1, 8, padding='same'), _rand_bn(r, 8),
                                                    dict(type='activation', fn='relu'),
                                                    _rand_conv(r, 3, 3, 8, 20, strides=(2, 2), padding='same', act='tanh'),
                                                    dict(type='maxpool', pool=(2, 2), strides=(2, 2), padding='valid'),
                                                    dict(type='flatten'), _rand_dense(r, 17 * ((-(-h // 2)) // 2) * 20, 2, 'softmax')],
    'bn_after_act_avgpool_global': lambda r, h: [_rand_conv(r, 5, 3, 1, 12, act='relu', bias=False), _rand_bn(r, 12),
                                                 dict(type='avgpool', pool=(3, 2), strides=(2, 2), padding='valid'),
                                                 _rand_conv(r, 1, 1, 12, 7, act='sigmoid'),
                                                 dict(type='globalavgpool'), _rand_dense(r, 7, 3, 'linear'),
                                                 dict(type='activation', fn='softmax')],
    'fused_pools_2x1_1x2_avg2x2_cin32': lambda r, h: [_rand_conv(r, 3, 3, 1, 32, act='relu'),
                                                      dict(type='maxpool', pool=(2, 1), strides=(2, 1), padding='valid'),
                                                      _rand_conv(r, 3, 3, 32, 64, act='relu'), _rand_bn(r, 64),
                                                      dict(type='maxpool', pool=(1, 2), strides=(1, 2), padding='valid'),
                                                      _rand_conv(r, 2, 2, 64, 32, padding='same', act='tanh'),
                                                      dict(type='avgpool', pool=(2, 2), strides=(2, 2), padding='valid'),
                                                      dict(type='flatten'),
                                                      _rand_dense(r, 15 * (((h - 2) - 2) // 2 // 2) * 32, 3, 'softmax')],
    'cin_not_multiple_of_4': lambda r, h: [_rand_conv(r, 2, 2, 1, 6, act='relu'), _rand_conv(r, 3, 2, 6, 10, act='relu'),
                                           dict(type='maxpool', pool=(3, 3), strides=(3, 3), padding='same'),
                                           dict(type='globalmaxpool'), _rand_dense(r, 10, 2, 'softmax')],
    # ZeroPadding2D merged into the convolutions behind it: asymmetric explicit padding on the first (PATCH) layer and on a 32-channel one
    'zeropadding2d_explicit': lambda r, h: [dict(_rand_conv(r, 4, 5, 1, 32, act='relu'), pad=(1, 2, 2, 1)),
                                            dict(type='maxpool', pool=(2, 2), strides=(2, 2), padding='valid'),
                                            dict(_rand_conv(r, 3, 3, 32, 32, act='relu'), pad=(2, 0, 0, 3)), _rand_bn(r, 32),
                                            dict(type='globalavgpool'), _rand_dense(r, 32, 3, 'softmax')],
    # activations beyond relu / sigmoid / tanh: ELU (layer, alpha 0.7), LeakyReLU (layer, alpha 0.2), selu and softplus (activation strings)
    'elu_leaky_selu_softplus': lambda r, h: [_rand_conv(r, 4, 5, 1, 32), dict(type='activation', fn='elu', alpha=0.7),
                                            dict(type='maxpool', pool=(2, 2), strides=(2, 2), padding='valid'),
                                            _rand_conv(r, 3, 3, 32, 32), _rand_bn(r, 32), dict(type='activation', fn='leaky_relu', alpha=0.2),
                                            _rand_conv(r, 3, 3, 32, 64, act='selu'), dict(type='globalavgpool'),
                                            _rand_dense(r, 64, 32, 'softplus'), _rand_dense(r, 32, 3, 'softmax')],
    # layers without a kernel of their own, lowered as ordinary convolutions on zero-filled kernels (keras_model.expand_generic_layers):
    # a dilated Conv2D, a DepthwiseConv2D (multiplier 2), the two halves of a SeparableConv2D, and ReLU(max_value) (ISS_OP_ACT code 8)
    'dilated_depthwise_separable_relu6': lambda r, h: [
        dict(_rand_conv(r, 3, 3, 1, 16, padding='same'), dilation=(2, 1)), dict(type='activation', fn='relu_max', alpha=1.5),
        dict(type='depthwise', W=r.normal(0, 0.4, (3, 3, 16, 2)).astype(np.float32), b=r.normal(0, 0.05, 32).astype(np.float32),
             strides=(1, 1), padding='valid', activation='relu', dilation=(1, 1)),
        dict(type='maxpool', pool=(2, 2), strides=(2, 2), padding='valid'),
        dict(type='depthwise', W=r.normal(0, 0.4, (3, 3, 32, 1)).astype(np.float32), b=None, strides=(1, 1), padding='same',
             activation='linear', dilation=(1, 2)),
        dict(type='activation', fn='relu_general', alpha=(0.1, 1.2, 0.15)),                  # keras.layers.ReLU in full (ISS_OP_ACT code 9)
        _rand_conv(r, 1, 1, 32, 32, act='relu'), dict(type='globalavgpool'), _rand_dense(r, 32, 3, 'softmax')],
    'standalone_bn_first': lambda r, h: [_rand_bn(r, 1), _rand_conv(r, 3, 3, 1, 4, act='relu'), dict(type='dropout'),
                                         dict(type='flatten'), _rand_dense(r, 66 * (h - 2) * 4, 64, 'relu'), _rand_bn(r, 64),
                                         _rand_dense(r, 64, 2, 'softmax')],
}


@pytest.mark.parametrize('name', sorted(TOPOLOGIES))
@pytest.mark.parametrize('nmel', [21, 24])
def test_layer_semantics(ctx, prec, name, nmel):
    rng = np.random.default_rng(sum(map(ord, name)) + nmel)
    layers = TOPOLOGIES[name](rng, nmel)
    comp = KM.compile_layers(layers, (68, nmel, 1))
    ctx.cnn_load(2, comp)
    mspec = _mspec(rng, 400)
    ctx.set_mspec(mspec)
    rows = rng.integers(0, 400 - 68, 301).astype(np.int32)
    probs, fin = ctx.cnn_probs(2, rows)
    ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
    assert np.array_equal(fin, rfin)
    err = np.abs(probs - ref).max()
    assert err < 1e-4, (name, err)


@pytest.mark.parametrize('net,nmel,ncls', [('smn', 21, 3), ('gender', 24, 2)])
def test_ina_like_net_on_real_features(ctx, prec, net, nmel, ncls):
    g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    mspec = g['musanmix_mspec']
    layers, shp = KM.synthetic_ina_like(nmel, ncls, seed=3)
    comp = KM.compile_layers(layers, shp)
    ctx.cnn_load(3, comp)
    assert ctx.cnn_flops(3) == comp.flops_per_sample <= ocnn.flops_per_sample(layers, shp)   # fused pools skip dropped rows
    ctx.set_mspec(mspec)
    rows = S._window_rows(len(mspec))
    probs, fin = ctx.cnn_probs(3, rows)
    patches, rfin = oseg.get_patches(mspec[:, :nmel].copy(), 68, 2)
    assert np.array_equal(fin, rfin)
    x = np.where(rfin[:, None, None], patches, 0)[..., None].astype(np.float32)
    ref = ocnn.forward(layers, x)
    ref[~rfin] = 0.5
    err = np.abs(probs - ref).max()
    print(f'{net} [{prec}]: max |p_gpu - p_oracle| = {err:.2e} over {len(rows)} slots')
    assert err < 1e-4
    # the unfused program (stand-alone pool kernels) must give the same answer
    ctx.cnn_load(3, KM.compile_layers(layers, shp, fuse_pool=False))
    probs2, _ = ctx.cnn_probs(3, rows)
    assert np.abs(probs2 - ref).max() < 1e-4 and np.abs(probs2 - probs).max() < 2e-5


def test_non_finite_and_constant_windows(ctx):
    rng = np.random.default_rng(11)
    layers, shp = KM.synthetic_ina_like(21, 3, seed=5)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    mspec = _mspec(rng, 300)
    mspec[100:110, 3] = -np.inf                  # digital silence -> log(0)
    mspec[200:268, :] = 1.25                     # constant window -> std 0 -> nan
    ctx.set_mspec(mspec)
    rows = np.arange(0, 300 - 68 + 1, dtype=np.int32)
    probs, fin = ctx.cnn_probs(3, rows)
    ref, rfin = _oracle_probs(layers, mspec, 21, rows)
    assert np.array_equal(fin, rfin) and (~fin).sum() > 70
    assert np.all(probs[~fin] == 0.5)
    assert np.abs(probs - ref).max() < PROB_TOL


def test_chunked_passes_agree(ctx):
    """Same slots through a tiny activation workspace (many passes) and a large one."""
    rng = np.random.default_rng(12)
    layers, shp = KM.synthetic_ina_like(24, 2, seed=6)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    mspec = _mspec(rng, 1000)
    ctx.set_mspec(mspec)
    rows = rng.integers(0, 1000 - 68, 1500).astype(np.int32)
    ctx.set_workspace_limit(64 << 20)
    a, _ = ctx.cnn_probs(3, rows)
    ctx.set_workspace_limit(6 << 30)
    b, _ = ctx.cnn_probs(3, rows)
    assert np.array_equal(a, b)
    c, _ = ctx.cnn_probs(3, rows[:1])          # a 1-slot pass may take another kernel variant (k order differs)
    assert np.abs(c[0] - a[0]).max() < 1e-6
    assert ctx.cnn_probs(3, rows[:0])[0].shape == (0, 2)


def test_generic_forward_nhwc(ctx, prec):
    rng = np.random.default_rng(13)
    layers = [_rand_conv(rng, 3, 3, 8, 16, padding='same', act='relu'), _rand_conv(rng, 1, 1, 16, 32, strides=(2, 2)),
              dict(type='flatten'), _rand_dense(rng, 5 * 6 * 32, 10)]
    comp = KM.compile_layers(layers, (9, 11, 8), patch_input=False)
    ctx.cnn_load(4, comp)
    x = rng.normal(0, 1, (37, 9, 11, 8)).astype(np.float32)
    out = ctx.cnn_forward(4, x)
    ref = ocnn.forward(layers, x)
    assert np.abs(out - ref).max() < 1e-4 * max(1, np.abs(ref).max())


def _conv_launches(ctx, fn):
    ctx.prof_enable(True)
    ctx.prof_reset()
    out = fn()
    n = ctx.prof_get(0)[1]
    ctx.prof_enable(False)
    return out, n


def _post_affine_net(nmel, nout, seed):
    """conv(relu) -> BatchNorm (a post-activation affine that cannot be folded into the conv) -> 5x3 conv."""
    r = np.random.default_rng(seed)
    return [_rand_conv(r, 4, 5, 1, 64, act='relu'), _rand_bn(r, 64), _rand_conv(r, 5, 3, 64, 64, act='relu'),
            dict(type='maxpool', pool=(4, 4), strides=(4, 4), padding='valid'), dict(type='flatten'),
            _rand_dense(r, 15 * ((nmel - 6) // 4) * 64, nout, 'softmax')], (68, nmel, 1)


@pytest.mark.parametrize('nmel,nout,kind', [(21, 3, 'ina'), (24, 2, 'ina'), (21, 3, 'post_affine')])
def test_shared_first_layer(ctx, nmel, nout, kind):
    """Overlapping windows (the segmenter's every-2nd-row list with edge replicas): the first conv runs once per
    log-mel row and the second conv normalises per window (ConvArgs::f_*).  Must agree with the per-window
    first layer (iss_set_diag ISS_DIAG_NO_SHARED_FIRST) and with the oracle; windows straddled by one LDS footprint, non-finite and
    constant windows included."""
    rng = np.random.default_rng(21)
    layers, shp = KM.synthetic_ina_like(nmel, nout, seed=9) if kind == 'ina' else _post_affine_net(nmel, nout, 9)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    mspec = _mspec(rng, 700)
    mspec[300:304, 2] = -np.inf
    mspec[500:568, :] = -2.5
    ctx.set_mspec(mspec)
    rows = np.concatenate([np.zeros(17, np.int32), np.arange(0, 700 - 68 + 1, 2, dtype=np.int32),
                           np.full(16, 632, np.int32)])
    (probs, fin), n_shared = _conv_launches(ctx, lambda: ctx.cnn_probs(3, rows))
    ctx.set_diag('no_shared_first')
    try:
        (probs_pw, fin_pw), n_pw = _conv_launches(ctx, lambda: ctx.cnn_probs(3, rows))
    finally:
        ctx.set_diag(0)
    assert n_shared < n_pw, 'the shared first layer did not run'      # one GEMM launch less per pass
    ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
    assert np.array_equal(fin, rfin) and np.array_equal(fin_pw, rfin) and 30 < (~fin).sum() < len(rows) - 100
    assert np.all(probs[~fin] == 0.5)
    print(f'shared vs per-window {np.abs(probs - probs_pw).max():.2e}, shared vs oracle {np.abs(probs - ref).max():.2e}, '
          f'per-window vs oracle {np.abs(probs_pw - ref).max():.2e}')
    # the per-window path rounds the NORMALISED input to 2^-16 (split bf16); the shared path's first layer is exact f32
    assert np.abs(probs - probs_pw).max() < 1e-4
    assert np.abs(probs - ref).max() < 1e-4


def test_shared_first_layer_needs_overlap(ctx):
    """Few scattered windows: the per-call rule (4-fold average overlap) keeps the per-window first layer, bit for bit."""
    rng = np.random.default_rng(22)
    layers, shp = KM.synthetic_ina_like(21, 3, seed=9)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    ctx.set_mspec(_mspec(rng, 3000))
    rows = np.sort(rng.integers(0, 3000 - 68, 40)).astype(np.int32)
    (a, _), n_a = _conv_launches(ctx, lambda: ctx.cnn_probs(3, rows))
    ctx.set_diag('no_shared_first')
    try:
        (b, _), n_b = _conv_launches(ctx, lambda: ctx.cnn_probs(3, rows))
    finally:
        ctx.set_diag(0)
    assert n_a == n_b and np.array_equal(a, b)


def _pw_program(rng, n, h, w, cin, specs):
    """A chain of 1x1 convolutions lowered by hand (no channel padding): specs = [(cout, act, residual?, stride, affine?)].
    A residual layer adds the activation two layers back IN PLACE (the ResNet bottleneck pattern, resnet.py:66-75).
    -> (CompiledNet, x, float64 reference of the flattened output)."""
    B = KM._Builder()
    x = rng.normal(0, 1, (n, h, w, cin)).astype(np.float32)
    acts = {0: lambda v: v, 1: lambda v: np.maximum(v, 0), 2: lambda v: 1 / (1 + np.exp(-v)), 3: np.tanh}
    shape, src = (h, w, cin), _native.BUF_INPUT
    vals = {src: x.astype(np.float64)}
    bufs = [0, 1, 2]
    hist = []
    for cout, act, res, stride, affine in specs:
        hh, ww, c = shape
        Wm = rng.normal(0, np.sqrt(1.0 / c), (cout, c)).astype(np.float32)
        b = rng.normal(0, 0.2, cout).astype(np.float32)
        ps = rng.uniform(0.5, 1.5, cout).astype(np.float32) if affine else None
        pt = rng.normal(0, 0.2, cout).astype(np.float32) if affine else None
        dst = hist[-2] if res else next(bb for bb in bufs if bb != src and (len(hist) < 1 or bb != hist[-1]))
        ho, wo = (hh - 1) // stride + 1, (ww - 1) // stride + 1
        xin = vals[src][:, ::stride, ::stride, :]
        y = xin @ Wm.astype(np.float64).T + b
        if res:
            y = y + vals[dst]
        y = acts[act](y)
        if affine:
            y = y * ps + pt
        shape = B.conv(src, dst, shape, Wm, 1, 1, stride, stride, 0, 0, ho, wo, bias=b, act=act, ps=ps, pt_=pt,
                       res=dst if res else -1)
        vals[dst] = y
        hist.append(dst)
        src = dst
    return B.finish((h, w, cin), int(np.prod(shape)), False), x, vals[src].reshape(n, -1)


@pytest.mark.parametrize('case', ['bottleneck_odd_channels', 'strided_projection', 'post_affine_and_sigmoid', 'k1024_n256', 'dense_192'])
def test_pointwise_streaming_kernels(ctx, prec, case):
    """conv_x3_pws_kernel / conv_x3_pws2_kernel (conv_pw.h): partial row tiles (M % 128 != 0), partial column tiles
    (Cout % 64 != 0), in-place residuals, the generic epilogue (sigmoid / tanh / post-activation affine), strided
    1x1 projections and deep K, against a float64 reference of the same GEMM chain."""
    rng = np.random.default_rng(sum(map(ord, case)))
    if case == 'bottleneck_odd_channels':       # M = 5 * 7 * 11 = 385: three full row tiles + one of a single row
        comp, x, ref = _pw_program(rng, 5, 7, 11, 64, [(96, 1, False, 1, False), (32, 1, False, 1, False),
                                                        (96, 1, True, 1, False), (100, 3, False, 1, False)])
    elif case == 'strided_projection':          # (8, 10) -> (4, 5): M = 9 * 20 = 180 rows spanning several samples per tile
        comp, x, ref = _pw_program(rng, 9, 8, 10, 64, [(128, 0, False, 2, False), (256, 1, False, 1, False),
                                                       (128, 1, False, 1, False), (256, 1, True, 1, False)])
    elif case == 'post_affine_and_sigmoid':
        comp, x, ref = _pw_program(rng, 3, 9, 15, 32, [(64, 1, False, 1, True), (128, 2, False, 1, True),
                                                       (64, 1, True, 1, True), (36, 0, False, 1, False)])
    elif case == 'dense_192':                   # the segmenter nets' dense head: 301 rows (64-row tiles, last one of 45), K = 4992 -> 192 -> 128 -> tanh 192
        comp, x, ref = _pw_program(rng, 301, 1, 1, 4992, [(192, 1, False, 1, False), (128, 1, False, 1, False),
                                                          (192, 3, False, 1, True)])
    else:                                       # 8 x 18 maps of ResNet-101's last stage: K = 1024 (32 k-steps), 4 x 128 columns
        comp, x, ref = _pw_program(rng, 4, 8, 18, 1024, [(256, 1, False, 1, False), (1024, 1, False, 1, False)])
    ctx.cnn_load(5, comp)
    out = ctx.cnn_forward(5, x)
    scale = max(1.0, np.abs(ref).max())
    err = np.abs(out - ref).max() / scale
    print(f'{case} [{prec}]: max rel err {err:.2e}')
    assert err < 1e-4, (case, err)


@pytest.mark.parametrize('nmel,nout', [(21, 3), (24, 2)])
def test_row_wise_first_layer_is_bit_identical(ctx, nmel, nout):
    """first_layer_rows_kernel (one thread per log-mel row and 4 channels, the input rows in registers) computes the same
    fmaf chains as first_layer_raw_kernel (one thread per output, ISS_DIAG_NO_FLROWS): the probabilities must not differ in a bit."""
    rng = np.random.default_rng(31)
    layers, shp = KM.synthetic_ina_like(nmel, nout, seed=4)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    ctx.set_mspec(rng.normal(-3, 2, (900, 24)).astype(np.float32))
    rows = np.arange(0, 900 - 68 + 1, 2, dtype=np.int32)
    outs = []
    for diag in (0, 'no_flrows'):
        ctx.set_diag(diag)
        try:
            p, f = ctx.cnn_probs(3, rows)
        finally:
            ctx.set_diag(0)
        outs.append(p)
    assert outs[0].shape == outs[1].shape == (417, nout)
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('nmel,nout', [(21, 3), (24, 2)])
def test_one_wave_per_simd_kernels_edge_sizes(ctx, nmel, nout):
    """conv_x3_wq_kernel / conv_x3_wq3_kernel (one wave per SIMD, two footprints, epilogue of a tile behind the next block's MFMAs)
    against the two-waves-per-SIMD kernels they replaced (iss_set_diag NO_WQ) and against the oracle, on window counts that hit
    the structure's edges: fewer groups than workgroups, an odd number of tiles (a group with one tile), a last tile with a few
    rows, one tile in all, 508-row tiles (17-column input), a window with non-finite values at a tile boundary.  Both kernel
    families accumulate the same products in the same order: the probabilities must be bit-identical."""
    rng = np.random.default_rng(41)
    layers, shp = KM.synthetic_ina_like(nmel, nout, seed=3)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    for T in (70, 72, 78, 100, 141, 300, 1000, 3001):
        mspec = _mspec(rng, T)
        if T >= 300:
            mspec[T // 2, 3] = np.inf
        ctx.set_mspec(mspec)
        rows = np.arange(0, T - 68 + 1, 2, dtype=np.int32)
        if len(rows) < 8:                                            # the shared first layer needs overlapping windows: pad the list
            rows = np.repeat(rows, 8)[:max(len(rows) * 4, 8)]
        ctx.prof_enable(True)
        ctx.prof_reset()
        p_new, f_new = ctx.cnn_probs(3, rows)
        used = {e['kernel'] for e in ctx.prof_instances()}
        ctx.prof_enable(False)
        ctx.set_diag('no_wq')
        try:
            p_old, f_old = ctx.cnn_probs(3, rows)
            # round 6: the 3x3 layers read the CHL layout their producers write (conv_x3_wq3h_kernel); 'no_hl' keeps f32 NHWC
            # between the layers (conv_x3_wq3_kernel) -- the same operand split, the same MFMA order: bit-identical too
            ctx.set_diag('no_hl')
            ctx.prof_enable(True)
            ctx.prof_reset()
            p_f32, f_f32 = ctx.cnn_probs(3, rows)
            used_f32 = {e['kernel'] for e in ctx.prof_instances()}
            ctx.prof_enable(False)
        finally:
            ctx.set_diag(0)
        assert np.array_equal(f_new, f_old) and np.array_equal(f_new, f_f32)
        ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
        assert np.array_equal(f_new, rfin)
        assert np.abs(p_new - ref).max() < 1e-4, (T, np.abs(p_new - ref).max())
        assert np.array_equal(p_new, p_old), (T, sorted(used), np.abs(p_new - p_old).max())
        assert np.array_equal(p_new, p_f32), (T, sorted(used), sorted(used_f32), np.abs(p_new - p_f32).max())
        if T >= 141:
            assert 'conv_x3_wq_kernel<5,3,true,false>' in used, (T, used)                          # <KH,KW,OUT_HL,F16>: conv2 writes CHL
            # conv3 (CHL out), conv4 (CHL of the flattened features out), the first dense layer on it
            assert {'conv_x3_wq3h_kernel<0,true,false>', 'conv_x3_wq3h_kernel<1,true,false>', 'conv_dhl_kernel<false,8>'} <= used, (T, used)
            assert sum(k.startswith('conv_x3_wq3_kernel') for k in used_f32) == 2 and 'conv_x3_wq_kernel<5,3,false,false>' in used_f32 \
                and not any('wq3h' in k or 'dhl' in k for k in used_f32) and 'conv_x3_pw_kernel<false>' in used_f32, (T, used_f32)
    # irregular window lists (what the VAD-gated gender pass hands over): gaps, runs, repeats -- a footprint then spans two
    # windows whose first rows are unrelated
    mspec = _mspec(rng, 6000)
    mspec[4000, 5] = np.nan
    ctx.set_mspec(mspec)
    for n, dense_runs in ((8, False), (333, False), (1531, True), (2500, True)):
        if dense_runs:                                               # runs of consecutive slots with gaps between them
            starts = np.sort(rng.choice(6000 - 68 - 40, n // 25 + 1, replace=False))
            rows = np.concatenate([np.arange(s, s + rng.integers(1, 40)) for s in starts])[:n].astype(np.int32)
        else:
            rows = np.sort(rng.integers(0, 6000 - 68 + 1, n)).astype(np.int32)
        p_new, f_new = ctx.cnn_probs(3, rows)
        ctx.set_diag('no_wq')
        try:
            p_old, f_old = ctx.cnn_probs(3, rows)
            ctx.set_diag('no_hl')
            p_f32, f_f32 = ctx.cnn_probs(3, rows)
        finally:
            ctx.set_diag(0)
        ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
        assert np.array_equal(f_new, rfin) and np.array_equal(f_new, f_old)
        assert np.abs(p_new - ref).max() < 1e-4, (n, np.abs(p_new - ref).max())
        assert np.array_equal(p_new, p_old), (n, np.abs(p_new - p_old).max())
        assert np.array_equal(p_new, p_f32), (n, np.abs(p_new - p_f32).max())


@pytest.mark.parametrize('net,nmel,ncls', [('smn', 21, 3), ('gender', 24, 2)])
def test_exact_f32_mode_takes_the_weight_stationary_kernels(ctx, net, nmel, ncls):
    """ISS_PREC_F32 on the stand-in nets over the segmenter's own overlapping window list: the three layers that carry the
    arithmetic run on the F32 form of the weight-stationary kernel (v_mfma_f32_32x32x2_f32 on the LDS footprint, shared first
    layer), the result matches the oracle to 1e-4 and the conv_igemm_kernel path it replaces (ISS_DIAG_NO_F32WS) to float32
    rounding of a different summation order, and the split-bf16 default to its own operand error."""
    rng = np.random.default_rng(31 + nmel)
    T = 1500
    mspec = _mspec(rng, T)
    mspec[400:403, 7] = -np.inf
    ctx.set_mspec(mspec)
    layers, shp = KM.synthetic_ina_like(nmel, ncls, seed=5)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    rows = S._window_rows(T)
    ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
    p_x3, f_x3 = ctx.cnn_probs(3, rows)
    ctx.set_precision(_native.PREC_F32)
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        p_f32, f_f32 = ctx.cnn_probs(3, rows)
        insts = [i['kernel'] for i in ctx.prof_instances()]
        ctx.prof_enable(False)
        ctx.set_diag('no_f32ws')
        p_old, f_old = ctx.cnn_probs(3, rows)
    finally:
        ctx.set_diag(0)
        ctx.set_precision(_native.PREC_BF16X3)
    assert sum(k.endswith('f32>') for k in insts) == 3, insts               # fused 5x3, 3x3 transposed, 3x3 pooled
    assert np.array_equal(f_f32, rfin) and np.array_equal(f_old, rfin) and np.array_equal(f_x3, rfin)
    e_or, e_old, e_x3 = np.abs(p_f32 - ref).max(), np.abs(p_f32 - p_old).max(), np.abs(p_f32 - p_x3).max()
    print(f'{net}: f32 ws vs oracle {e_or:.2e}, vs conv_igemm_kernel {e_old:.2e}, vs bf16x3 {e_x3:.2e}')
    assert e_or < 1e-4 and e_old < 2e-5 and e_x3 < 1e-4


def test_precision_guard_escalates_a_net_with_inflated_activation_range(ctx):
    """Round 6 (include/iss.h, iss_set_precision_guard): the first iss_cnn_probs call of a patch network in split-bf16 mode compares
    both arithmetic modes on up to 256 of its own windows.  The calibrated stand-in passes (max |d log p| a few 1e-4 at most, mode
    kept); the same network with its last two layers scaled so that its logits are ~6 x larger -- what confident real weights
    look like -- trips the 5e-4 threshold, is switched to exact f32 by the library, and its results then sit within the north
    star's 1e-3 of the oracle's log-probabilities, which the split-bf16 results (guard off) do not."""
    import bench
    nmel, ncls = 21, 3
    pcm = bench.synth_recording(0, 60 * 16000, 'cpu').numpy()            # the bench generator's audio: silence, noise, a voiced source, chords
    ctx.set_signal(pcm)
    T = ctx.sidekit()
    mspec = ctx.get_mspec()
    rows = S._window_rows(T)
    layers, shp = KM.synthetic_ina_like(nmel, ncls, seed=1)
    ctx.set_precision_guard(5e-4)                                        # (the shared test context runs with the guard off)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    assert ctx.cnn_precision_info(3)['state'] == 'pending'
    p0, f0 = ctx.cnn_probs(3, rows)
    info = ctx.cnn_precision_info(3)
    print('stand-in:', info)
    assert info['state'] == 'passed' and info['mode'] == 'bf16x3' and 0 <= info['max_dlogp'] < 5e-4 and info['slots'] > 100, info
    ref0, _ = _oracle_probs(layers, mspec, nmel, rows)
    assert np.abs(p0 - ref0).max() < 1e-4, np.abs(p0 - ref0).max()

    hot = [dict(L) for L in layers]
    for i in (-2, -1):                                                   # dense(192, 128, relu), dense(128, ncls, softmax)
        hot[i]['W'] = (hot[i]['W'] * 2.5).astype(np.float32)
        hot[i]['b'] = (hot[i]['b'] * 2.5).astype(np.float32)
    ref, fin = _oracle_probs(hot, mspec, nmel, rows)
    ok = fin[:, None] & (ref > 1e-30)

    def dlogp(p):
        with np.errstate(divide='ignore'):
            return np.abs(np.log(p.astype(np.float64)) - np.log(ref.astype(np.float64)))[ok & (p > 1e-30)].max()

    ctx.set_precision_guard(0)                                           # guard off: split-bf16 on a net it is too coarse for
    try:
        ctx.cnn_load(3, KM.compile_layers(hot, shp))
        p_x3, _ = ctx.cnn_probs(3, rows)
        assert ctx.cnn_precision_info(3)['state'] == 'pending'
    finally:
        ctx.set_precision_guard(5e-4)
    ctx.cnn_load(3, KM.compile_layers(hot, shp))                         # guard on (the library's default)
    ctx.prof_enable(True)
    ctx.prof_reset()
    p_g, f_g = ctx.cnn_probs(3, rows)
    ctx.prof_enable(False)
    info = ctx.cnn_precision_info(3)
    print('inflated:', info, 'split-bf16 max |d log p| vs oracle', dlogp(p_x3), 'guarded', dlogp(p_g))
    # split bf16 fails the probe; fp16 halves (the same speed) pass it: that is the mode the network now runs in
    assert info['state'] == 'escalated' and info['mode'] == 'f16x3' and info['max_dlogp'] > 5e-4, info
    used = {e['kernel'] for e in ctx.prof_instances()}
    assert _F16_KERNELS & used, used
    assert np.array_equal(f_g, fin)
    assert dlogp(p_g) < 1e-3, dlogp(p_g)
    assert dlogp(p_x3) > 1e-3 and dlogp(p_x3) > 3 * dlogp(p_g) and info['max_dlogp_in_use'] < 5e-4
    # a threshold neither split mode meets: exact f32
    ctx.set_precision_guard(2e-6)
    ctx.cnn_load(3, KM.compile_layers(hot, shp))
    p_e, _ = ctx.cnn_probs(3, rows)
    ctx.set_precision_guard(5e-4)
    info_e = ctx.cnn_precision_info(3)
    print('threshold 2e-6:', info_e, 'max |d log p| vs oracle', dlogp(p_e))
    assert info_e['state'] == 'escalated' and info_e['mode'] == 'f32', info_e
    assert dlogp(p_e) < 1e-3
    ctx.cnn_load(3, KM.compile_layers(hot, shp))
    p_g2, _ = ctx.cnn_probs(3, rows)
    assert np.array_equal(p_g2, p_g)
    p_again, _ = ctx.cnn_probs(3, rows)                                  # decided once: no second probe, same arithmetic
    assert np.array_equal(p_again, p_g) and ctx.cnn_precision_info(3)['slots'] == info['slots']
    # the caller's own choice wins and is never probed
    ctx.cnn_load(3, KM.compile_layers(hot, shp))
    ctx.cnn_set_net_precision(3, _native.PREC_BF16X3)
    p_fix, _ = ctx.cnn_probs(3, rows)
    assert ctx.cnn_precision_info(3)['state'] == 'fixed' and np.array_equal(p_fix, p_x3)
    ctx.set_precision_guard(0)


@pytest.mark.parametrize('net,nmel,ncls', [('smn', 21, 3), ('gender', 24, 2)])
def test_f16x3_mode(ctx, net, nmel, ncls):
    """ISS_PREC_F16X3 (round 6): fp16 instead of bf16 operand halves in the kernels that carry the arithmetic (conv2 with its CHL
    output, conv3 / conv4 on it, the long-K dense layer), exact f32 for the small trailing layers.  On the calibrated stand-ins
    over the bench generator's audio: the same windows, finite masks and arg-max as the oracle, and the log-probabilities several
    times closer to it than the split-bf16 default's (tests/precision_emulation.py: 10-15 x less operand error)."""
    import bench
    pcm = bench.synth_recording(1, 60 * 16000, 'cpu').numpy()
    ctx.set_signal(pcm)
    T = ctx.sidekit()
    mspec = ctx.get_mspec()
    rows = S._window_rows(T)
    layers, shp = KM.synthetic_ina_like(nmel, ncls, seed=1 if net == 'smn' else 2)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    ref, rfin = _oracle_probs(layers, mspec, nmel, rows)
    p_b, f_b = ctx.cnn_probs(3, rows)
    ctx.set_precision(_native.PREC_F16X3)
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        p_h, f_h = ctx.cnn_probs(3, rows)
        used = sorted({e['kernel'] for e in ctx.prof_instances()})
        ctx.prof_enable(False)
    finally:
        ctx.set_precision(_native.PREC_BF16X3)
    assert np.array_equal(f_h, rfin) and np.array_equal(f_b, rfin)
    ok = rfin[:, None] & (ref > 1e-30)

    def dlogp(p):
        with np.errstate(divide='ignore'):
            return np.abs(np.log(p.astype(np.float64)) - np.log(ref.astype(np.float64)))[ok & (p > 1e-30)].max()
    print(f'{net}: max |d log p| vs oracle: bf16x3 {dlogp(p_b):.2e}, f16x3 {dlogp(p_h):.2e}; kernels {used}')
    assert _F16_KERNELS <= set(used), used                                   # conv2, conv3, conv4, dense
    assert any(k.startswith('conv_igemm_kernel') for k in used), used        # the small trailing layers: exact f32
    assert np.abs(p_h - ref).max() < 1e-4 and np.array_equal(p_h.argmax(1)[rfin], ref.argmax(1)[rfin])
    assert dlogp(p_h) < 1e-4 and dlogp(p_h) < 0.5 * dlogp(p_b)


@pytest.mark.parametrize('width', [36, 196, 320, 512, 772])
def test_first_dense_layer_wider_than_one_column_tile(ctx, prec, width):
    """conv_dhl_kernel holds a 128 x 192 tile; a wider first dense layer runs as column tiles (blockIdx.y, its packed weights one
    [K / 32][4][2][192][8] block per tile, zero columns behind Cout).  Widths below, at a ragged multiple of and far above 192:
    within 1e-4 of the oracle, and bit-identical to conv_x3_pw_kernel on the f32 hand-over (ISS_DIAG 'no_hl': the same operand
    split, the same products in the same k order)."""
    import topologies as TP
    rng = np.random.default_rng(500 + width)
    spec = TP.SPECS['standin'][:10] + [('flatten',), ('dense', width, 'linear'), ('bn_relu',), ('drop',), ('dense', 128)]
    layers, shp = TP.build(spec, 21, 3, seed=width)
    ctx.cnn_load(3, KM.compile_layers(layers, shp))
    for T in (141, 700, 2400):
        mspec = _mspec(rng, T)
        ctx.set_mspec(mspec)
        rows = np.arange(0, T - 68 + 1, 2, dtype=np.int32)
        ctx.prof_enable(True)
        ctx.prof_reset()
        p_new, f_new = ctx.cnn_probs(3, rows)
        used = {e['kernel'] for e in ctx.prof_instances()}
        ctx.prof_enable(False)
        ctx.set_diag('no_hl')
        try:
            ctx.prof_enable(True)
            ctx.prof_reset()
            p_f32, f_f32 = ctx.cnn_probs(3, rows)
            used_f32 = {e['kernel'] for e in ctx.prof_instances()}
            ctx.prof_enable(False)
        finally:
            ctx.set_diag(0)
        ref, rfin = _oracle_probs(layers, mspec, 21, rows)
        assert np.array_equal(f_new, rfin) and np.array_equal(f_new, f_f32)
        assert np.abs(p_new - ref).max() < 1e-4, (width, T, np.abs(p_new - ref).max())
        if prec != 'f32':
            assert any(k.startswith('conv_dhl_kernel') for k in used), (width, T, sorted(used))
            # (with fp16 halves the f32 hand-over runs conv2 .. conv4 on bf16 halves -- the fp16 forms exist on the CHL path only: nothing to equal)
            if prec == 'bf16x3':
                assert any(k.startswith('conv_x3_pw_kernel') for k in used_f32), (width, T, sorted(used_f32))
                assert np.array_equal(p_new, p_f32), (width, T, np.abs(p_new - p_f32).max())
