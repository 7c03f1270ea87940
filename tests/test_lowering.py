"""CPU: the Keras-layer-list -> op-program lowering (keras_model.compile_layers), executed by tests/prog_interp.py
(torch-CPU float64), against the Keras-semantics oracle -- every sweep topology, including the channel-padded ones."""
import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, _native as N
from oracle import keras_cnn as ocnn
import prog_interp
import topologies as TP


@pytest.mark.parametrize('name', sorted(TP.SPECS))
def test_lowered_program_equals_oracle(name):
    rng = np.random.default_rng(3)
    for net, (layers, shp) in TP.nets(name).items():
        x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
        comp = KM.compile_layers(layers, shp)
        got = prog_interp.run(comp, x)
        want = ocnn.forward(layers, x)
        assert got.shape == want.shape and np.abs(got - want).max() < 2e-5, (name, net, np.abs(got - want).max())
        # algorithmic flops ignore channel padding (fused 'valid' pools may drop an odd last row / column)
        assert KM.compile_layers(layers, shp, fuse_pool=False).flops_per_sample == ocnn.flops_per_sample(layers, shp)
        assert comp.flops_per_sample <= ocnn.flops_per_sample(layers, shp)
        # every conv behind the first layer reads a multiple of 32 channels (vectorised MFMA kernels)
        for R in comp.prog[1:]:
            if R[N.C_OP] == N.OP_CONV:
                assert R[N.C_CIN] % 32 == 0, (name, net, R[N.C_CIN])


def test_channel_padding_is_transparent():
    layers, shp = TP.build(TP.SPECS['ch48_96'], 21, 3, 5)
    a = KM.compile_layers(layers, shp, pad_channels=True)
    b = KM.compile_layers(layers, shp, pad_channels=False)
    assert [int(r[N.C_COUT]) for r in a.prog if r[N.C_OP] == N.OP_CONV][:4] == [64, 64, 96, 96]
    assert [int(r[N.C_COUT]) for r in b.prog if r[N.C_OP] == N.OP_CONV][:4] == [48, 48, 96, 96]
    x = np.random.default_rng(0).normal(0, 1, (2,) + shp).astype(np.float32)
    assert np.abs(prog_interp.run(a, x) - prog_interp.run(b, x)).max() < 1e-12
    assert a.flops_per_sample == b.flops_per_sample and a.out_dim == b.out_dim == 3


def test_resnet101_lowering_equals_oracle():
    """keras_model.compile_resnet101 (BatchNorm folding, residual wiring with in-place adds, stride-2 shortcuts, statistics
    pooling order, embedding layer) executed on CPU against the torch restatement of resnet.py:115-130 (oracle/vbx.py)."""
    from oracle import vbx as ovbx
    params = KM.synthetic_resnet101(3)
    comp = KM.compile_resnet101(params)
    x = np.random.default_rng(5).normal(0, 1, (2, 64, 144)).astype(np.float32)
    ref = ovbx.resnet101_forward(params, x)
    got = prog_interp.run(comp, x[..., None])
    assert got.shape == ref.shape == (2, 256)
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())



def _post_affine_rows(comp):
    return int(sum(1 for R in comp.prog if R[N.C_OP] == N.OP_CONV and R[N.C_PSOFF] >= 0))


def test_post_activation_batchnorm_is_folded_forward_where_that_is_exact():
    """`Conv2D(activation='relu')` + `BatchNormalization()` (BatchNorm BEHIND the activation, a common Keras idiom): the affine map
    moves into the next Dense / unpadded Conv2D (through dropout, flatten, average pools, and max pools when every scale is
    positive), so the program has no post-activation affine left and looks like a conv - BN - relu net to the kernels; it stays an
    epilogue affine in front of a zero-padded convolution and, with a negative scale, in front of a max pool."""
    rng = np.random.default_rng(11)
    for net, (layers, shp) in TP.nets('relu_then_bn').items():
        a = KM.compile_layers(layers, shp)
        b = KM.compile_layers(layers, shp, fold_post_bn=False)
        assert _post_affine_rows(a) == 0 and _post_affine_rows(b) == 4, (net, _post_affine_rows(a), _post_affine_rows(b))
        x = rng.normal(0, 1, (4,) + shp).astype(np.float32)
        want = ocnn.forward(layers, x)
        assert np.abs(prog_interp.run(a, x) - want).max() < 2e-5 and np.abs(prog_interp.run(b, x) - want).max() < 2e-5
        assert a.flops_per_sample == b.flops_per_sample
    # guards: 'same' consumer, negative scale in front of a max pool, average pool (any sign), global max pool, tanh activation
    spec = [('conv', 4, 5, 32), ('relu_bn',), ('conv', 3, 3, 32, 'same'), ('relu_bn',), ('maxpool', 2, 2), ('conv', 3, 3, 64), ('relu_bn',),
            ('avgpool', 2, 2), ('conv', 3, 3, 64), ('relu_bn',), ('gap',), ('dense', 32, 'tanh')]
    layers, shp = TP.build(spec, 21, 3, 5)
    bns = [L for L in layers if L['type'] == 'batchnorm']
    bns[1]['gamma'][3] = -0.7                                     # in front of the max pool: not foldable
    bns[2]['gamma'][5] = -0.4                                     # in front of the average pool: foldable all the same
    comp = KM.compile_layers(layers, shp)
    assert _post_affine_rows(comp) == 2                           # conv1 (its consumer zero-pads) and conv2 (negative scale, max pool)
    x = rng.normal(0, 1, (4,) + shp).astype(np.float32)
    assert np.abs(prog_interp.run(comp, x) - ocnn.forward(layers, x)).max() < 2e-5


def test_zeropadding2d_is_merged_into_the_convolution_behind_it():
    """Keras `ZeroPadding2D(((1, 2), (2, 1)))` + `Conv2D(padding='valid')`: parsed into one conv row with explicit (asymmetric)
    padding; anything else behind a ZeroPadding2D is refused."""
    rng = np.random.default_rng(4)
    k1 = rng.normal(0, 0.3, (4, 5, 1, 8)).astype(np.float32)
    k2 = rng.normal(0, 0.2, (3, 3, 8, 16)).astype(np.float32)
    d = rng.normal(0, 0.1, (16, 3)).astype(np.float32)
    cfg = {'class_name': 'Sequential', 'config': {'layers': [
        {'class_name': 'ZeroPadding2D', 'config': {'name': 'zp1', 'padding': [[1, 2], [2, 1]], 'batch_input_shape': [None, 68, 21, 1]}},
        {'class_name': 'Conv2D', 'config': {'name': 'c1', 'padding': 'valid', 'activation': 'relu', 'strides': [1, 1]}},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'p1', 'pool_size': [2, 2]}},
        {'class_name': 'ZeroPadding2D', 'config': {'name': 'zp2', 'padding': 1}},
        {'class_name': 'Conv2D', 'config': {'name': 'c2', 'padding': 'valid', 'activation': 'tanh'}},
        {'class_name': 'GlobalAveragePooling2D', 'config': {'name': 'g'}},
        {'class_name': 'Dense', 'config': {'name': 'd', 'activation': 'softmax'}}]}}
    weights = {'c1': {'kernel': k1, 'bias': np.zeros(8, np.float32)}, 'c2': {'kernel': k2, 'bias': rng.normal(0, 0.1, 16).astype(np.float32)},
               'd': {'kernel': d, 'bias': np.zeros(3, np.float32)}}
    layers, shp = KM.layers_from_keras_config(cfg, weights)
    assert shp == (68, 21, 1) and [L['type'] for L in layers] == ['conv2d', 'maxpool', 'conv2d', 'globalavgpool', 'dense']
    assert layers[0]['pad'] == (1, 2, 2, 1) and layers[2]['pad'] == (1, 1, 1, 1)
    comp = KM.compile_layers(layers, shp)
    rows = [R for R in comp.prog if R[N.C_OP] == N.OP_CONV]
    assert (int(rows[0][N.C_HO]), int(rows[0][N.C_WO]), int(rows[0][N.C_PT]), int(rows[0][N.C_PL])) == (68 + 3 - 4 + 1, 21 + 3 - 5 + 1, 1, 2)
    x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
    want = ocnn.forward(layers, x)
    assert np.abs(prog_interp.run(comp, x) - want).max() < 2e-5 and np.abs(ocnn.forward_naive(layers, x[:1]) - want[:1]).max() < 1e-5
    assert comp.flops_per_sample <= ocnn.flops_per_sample(layers, shp)
    bad = {'class_name': 'Sequential', 'config': {'layers': [cfg['config']['layers'][0], cfg['config']['layers'][2]]}}
    with pytest.raises(NotImplementedError):
        KM.layers_from_keras_config(bad, weights)


def test_more_activations_parse_lower_and_run():
    """ELU / LeakyReLU layers (with their alpha), `ReLU(negative_slope=...)`, and 'selu' / 'softplus' / 'elu' activation strings: parsed,
    lowered onto ISS_OP_ACT rows (codes 4..7, the parameter in ISS_C_ACTPARAM) behind linear producers, and executed like the oracle's
    Keras semantics."""
    rng = np.random.default_rng(9)
    k1 = rng.normal(0, 0.3, (4, 5, 1, 8)).astype(np.float32)
    k2 = rng.normal(0, 0.2, (3, 3, 8, 16)).astype(np.float32)
    d1 = rng.normal(0, 0.2, (16, 12)).astype(np.float32)
    d2 = rng.normal(0, 0.2, (12, 3)).astype(np.float32)
    cfg = {'class_name': 'Sequential', 'config': {'layers': [
        {'class_name': 'Conv2D', 'config': {'name': 'c1', 'padding': 'valid', 'activation': 'linear', 'batch_input_shape': [None, 68, 21, 1]}},
        {'class_name': 'LeakyReLU', 'config': {'name': 'l1', 'alpha': 0.15}},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'p1', 'pool_size': [2, 2]}},
        {'class_name': 'Conv2D', 'config': {'name': 'c2', 'padding': 'same', 'activation': 'selu'}},
        {'class_name': 'ELU', 'config': {'name': 'e1', 'alpha': 0.5}},
        {'class_name': 'GlobalMaxPooling2D', 'config': {'name': 'g'}},
        {'class_name': 'Dense', 'config': {'name': 'd1', 'activation': 'softplus'}},
        {'class_name': 'ReLU', 'config': {'name': 'r1', 'negative_slope': 0.05}},
        {'class_name': 'ReLU', 'config': {'name': 'r2', 'negative_slope': 0.1, 'max_value': 0.9, 'threshold': 0.2}},
        {'class_name': 'Dense', 'config': {'name': 'd2', 'activation': 'softmax'}}]}}
    weights = {'c1': {'kernel': k1, 'bias': rng.normal(0, 0.1, 8).astype(np.float32)}, 'c2': {'kernel': k2, 'bias': np.zeros(16, np.float32)},
               'd1': {'kernel': d1, 'bias': np.zeros(12, np.float32)}, 'd2': {'kernel': d2, 'bias': np.zeros(3, np.float32)}}
    layers, shp = KM.layers_from_keras_config(cfg, weights)
    fns = [(L['fn'], L.get('alpha')) for L in layers if L['type'] == 'activation']
    assert fns == [('leaky_relu', 0.15), ('elu', 0.5), ('leaky_relu', 0.05), ('relu_general', (0.1, 0.9, 0.2))]
    comp = KM.compile_layers(layers, shp)
    acts = [(int(R[N.C_ACT]), float(np.array([int(R[N.C_ACTPARAM])], np.int32).view(np.float32)[0])) for R in comp.prog if R[N.C_OP] == N.OP_ACT]
    assert [a for a, _ in acts] == [5, 6, 4, 7, 5, 9] and abs(acts[0][1] - 0.15) < 1e-7 and abs(acts[2][1] - 0.5) < 1e-7 and acts[1][1] == 0.0
    assert all(int(R[N.C_ACT]) == 0 for R in comp.prog if R[N.C_OP] == N.OP_CONV)           # the producers run linear
    first = [R for R in comp.prog if R[N.C_OP] == N.OP_CONV][0]
    assert int(first[N.C_FPOOLH]) == 2 and int(first[N.C_FPOOLW]) == 2                        # max pool fused IN FRONT of the leaky relu
    x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
    want = ocnn.forward(layers, x)
    assert np.abs(prog_interp.run(comp, x) - want).max() < 2e-5 and np.abs(ocnn.forward_naive(layers, x[:1]) - want[:1]).max() < 2e-5
    thr, _ = KM.layers_from_keras_config({'class_name': 'Sequential', 'config': {'layers': [
        {'class_name': 'ReLU', 'config': {'name': 'r', 'threshold': 0.5, 'batch_input_shape': [None, 68, 21, 1]}}]}}, {})
    assert thr[0]['fn'] == 'relu_general' and thr[0]['alpha'] == (0.0, float('inf'), 0.5)


def test_layers_without_a_kernel_of_their_own_are_lowered_through_ordinary_convolutions():
    """`keras.models.load_model` (segmenter.py:129-131) takes any model_config.  Dilated Conv2D, DepthwiseConv2D, SeparableConv2D and
    ReLU(max_value=...) have no dedicated kernel: they are parsed from the Keras config and lowered as ordinary convolutions on
    zero-filled kernels (keras_model.expand_generic_layers) / an ISS_OP_ACT row, and compute what the oracle's independent torch
    semantics (dilation / groups arguments) compute.  The string activation 'leaky_relu' is keras.activations.leaky_relu (slope 0.2),
    not the LeakyReLU layer (0.3)."""
    rng = np.random.default_rng(12)
    nrm = lambda *sh, s=0.3: rng.normal(0, s, sh).astype(np.float32)
    cfg = {'class_name': 'Sequential', 'config': {'layers': [
        {'class_name': 'Conv2D', 'config': {'name': 'c1', 'padding': 'same', 'activation': 'linear', 'dilation_rate': [2, 3],
                                            'batch_input_shape': [None, 68, 21, 1]}},
        {'class_name': 'ReLU', 'config': {'name': 'r6', 'max_value': 0.8}},
        {'class_name': 'DepthwiseConv2D', 'config': {'name': 'dw', 'padding': 'valid', 'depth_multiplier': 2, 'activation': 'leaky_relu',
                                                     'strides': [2, 1]}},
        {'class_name': 'SeparableConv2D', 'config': {'name': 'sp', 'padding': 'same', 'activation': 'relu', 'dilation_rate': 2}},
        {'class_name': 'GlobalAveragePooling2D', 'config': {'name': 'g'}},
        {'class_name': 'Dense', 'config': {'name': 'd', 'activation': 'softmax'}}]}}
    weights = {'c1': {'kernel': nrm(3, 2, 1, 8), 'bias': nrm(8, s=0.1)},
               'dw': {'depthwise_kernel': nrm(3, 3, 8, 2), 'bias': nrm(16, s=0.1)},
               'sp': {'depthwise_kernel': nrm(3, 3, 16, 1), 'pointwise_kernel': nrm(1, 1, 16, 24), 'bias': nrm(24, s=0.1)},
               'd': {'kernel': nrm(24, 3), 'bias': np.zeros(3, np.float32)}}
    layers, shp = KM.layers_from_keras_config(cfg, weights)
    assert [L['type'] for L in layers] == ['conv2d', 'activation', 'depthwise', 'depthwise', 'conv2d', 'globalavgpool', 'dense']
    assert layers[0]['dilation'] == (2, 3) and layers[1] == dict(type='activation', name='r6', fn='relu_max', alpha=0.8)
    assert layers[2]['alpha'] == 0.2 and layers[2]['strides'] == (2, 1) and layers[3]['b'] is None and layers[3]['dilation'] == (2, 2)
    ex = KM.expand_generic_layers(layers)
    assert ex[0]['W'].shape == (5, 4, 1, 8) and np.count_nonzero(ex[0]['W']) <= 3 * 2 * 8                 # (kh - 1) dy + 1, (kw - 1) dx + 1
    assert ex[2]['type'] == 'conv2d' and ex[2]['W'].shape == (3, 3, 8, 16) and ex[3]['W'].shape == (5, 5, 16, 16)
    assert all(ex[2]['W'][:, :, c, o].any() == (o // 2 == c) for c in range(8) for o in range(16))         # output c * m + j <- input c
    comp = KM.compile_layers(layers, shp)
    assert any(int(R[N.C_OP]) == N.OP_ACT and int(R[N.C_ACT]) == 8 for R in comp.prog)
    x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
    want = ocnn.forward(layers, x)
    got = prog_interp.run(comp, x)
    assert np.abs(got - want).max() < 2e-5, np.abs(got - want).max()
    # the model's own MACs, not the zero-filled kernels': dilation and depthwise expansion do not inflate the count
    dense_equiv = KM.compile_layers(ex, shp)
    assert comp.flops_per_sample < 0.5 * sum(2.0 * np.prod(L['W'].shape) * 68 * 21 for L in ex if L['type'] == 'conv2d')
    assert dense_equiv.flops_per_sample == comp.flops_per_sample
    # a depthwise convolution by hand: channel c of the input through filter (c, m) only
    one = [dict(type='depthwise', W=nrm(2, 2, 3, 2), b=None, strides=(1, 1), padding='valid', activation='linear', dilation=(1, 1))]
    xi = rng.normal(0, 1, (1, 4, 4, 3)).astype(np.float32)
    y = ocnn.forward(one + [dict(type='flatten')], xi).reshape(3, 3, 6)
    for c in range(3):
        for m in range(2):
            ref = sum(xi[0, dy:dy + 3, dx:dx + 3, c] * one[0]['W'][dy, dx, c, m] for dy in range(2) for dx in range(2))
            assert np.abs(y[:, :, c * 2 + m] - ref).max() < 1e-6


# ------------------------------------------------------------------------------ graph-shaped models (ISS_OP_ELT rows)
import graph_nets as GN


@pytest.mark.parametrize('name', sorted(GN.NETS))
def test_graph_shaped_models_lower_and_run(name):
    """Functional models that are not a chain (residual adds, concatenations, permutes / reshapes, several readers of a tensor):
    the lowered program -- chains through the usual fusions, merges as ISS_OP_ELT rows, buffers handed out by liveness -- computes
    what the Keras-semantics oracle computes, and the two oracle implementations agree with each other."""
    rng = np.random.default_rng(11)
    for nmel, ncls in ((21, 3), (24, 2)):
        layers, shp = GN.NETS[name](nmel, ncls, 5)
        x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
        comp = KM.compile_layers(layers, shp)
        got = prog_interp.run(comp, x)
        want = ocnn.forward(layers, x)
        assert got.shape == want.shape == (3, ncls) and np.isfinite(got).all()
        assert np.abs(got - want).max() < 2e-5, (name, nmel, np.abs(got - want).max())
        assert any(R[N.C_OP] == N.OP_ELT for R in comp.prog) or name == 'permute_reshape' and any(R[N.C_ACT] == N.ELT_PERMUTE for R in comp.prog)
        # no row overwrites a buffer somebody still has to read (the interpreter would have computed garbage), and the count of
        # buffers stays small
        assert len(comp.buf_elems) <= 6, len(comp.buf_elems)
        comp2 = KM.compile_layers(layers, shp, pad_channels=False, fuse_pool=False, fold_post_bn=False)
        assert np.abs(prog_interp.run(comp2, x) - want).max() < 2e-5
    small, shp = GN.NETS[name](21, 3, 9)
    xs = rng.normal(0, 1, (1,) + shp).astype(np.float32)
    assert np.abs(ocnn.forward(small, xs) - ocnn.forward_naive(small, xs)).max() < 2e-5


@pytest.mark.parametrize('seed', range(12))
def test_random_graphs_lower_and_run(seed):
    rng = np.random.default_rng(seed)
    layers, shp = GN.random_graph(seed, 21 if seed % 2 else 24, 3 if seed % 2 else 2)
    x = rng.normal(0, 1, (2,) + shp).astype(np.float32)
    got = prog_interp.run(KM.compile_layers(layers, shp), x)
    want = ocnn.forward(layers, x)
    assert np.isfinite(got).all() and np.abs(got - want).max() < 2e-5, (seed, np.abs(got - want).max())


def _keras_functional_config(fmt):
    """A small residual + concatenate model as `model.to_json()` writes it: Keras 2 ('inbound_nodes': [[[name, 0, 0, {}], ...]]) or
    Keras 3 ('inbound_nodes': [{'args': [...__keras_tensor__...], 'kwargs': {}}])."""
    def inb(*names):
        if fmt == 2:
            return [[[nm, 0, 0, {}] for nm in names]]
        kt = [{'class_name': '__keras_tensor__', 'config': {'shape': [None, 1], 'dtype': 'float32', 'keras_history': [nm, 0, 0]}} for nm in names]
        return [{'args': [kt[0]] if len(kt) == 1 else [kt], 'kwargs': {}}]
    L = [
        {'class_name': 'InputLayer', 'config': {'name': 'in', **({'batch_input_shape': [None, 68, 21, 1]} if fmt == 2 else {'batch_shape': [None, 68, 21, 1]})}, 'inbound_nodes': []},
        {'class_name': 'Conv2D', 'config': {'name': 'c1', 'filters': 8, 'kernel_size': [3, 3], 'padding': 'same', 'activation': 'relu'}, 'inbound_nodes': inb('in')},
        {'class_name': 'ZeroPadding2D', 'config': {'name': 'zp', 'padding': [[1, 1], [1, 1]]}, 'inbound_nodes': inb('c1')},
        {'class_name': 'Conv2D', 'config': {'name': 'c2', 'filters': 8, 'kernel_size': [3, 3], 'padding': 'valid', 'activation': 'linear'}, 'inbound_nodes': inb('zp')},
        {'class_name': 'BatchNormalization', 'config': {'name': 'bn', 'axis': 3, 'epsilon': 1e-3}, 'inbound_nodes': inb('c2')},
        {'class_name': 'Add', 'config': {'name': 'add'}, 'inbound_nodes': inb('c1', 'bn')},
        {'class_name': 'SeparableConv2D', 'config': {'name': 'sep', 'filters': 8, 'kernel_size': [3, 3], 'padding': 'same', 'activation': 'relu'}, 'inbound_nodes': inb('add')},
        {'class_name': 'Concatenate', 'config': {'name': 'cat', 'axis': -1}, 'inbound_nodes': inb('add', 'sep')},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'mp', 'pool_size': [4, 4]}, 'inbound_nodes': inb('cat')},
        {'class_name': 'Permute', 'config': {'name': 'pm', 'dims': [2, 1, 3]}, 'inbound_nodes': inb('mp')},
        {'class_name': 'Flatten', 'config': {'name': 'fl'}, 'inbound_nodes': inb('pm')},
        {'class_name': 'Dense', 'config': {'name': 'out', 'units': 3, 'activation': 'softmax'}, 'inbound_nodes': inb('fl')},
    ]
    return {'class_name': 'Functional' if fmt == 3 else 'Model',
            'config': {'name': 'm', 'layers': L, 'input_layers': [['in', 0, 0]], 'output_layers': [['out', 0, 0]]}}


@pytest.mark.parametrize('fmt', (2, 3))
def test_functional_model_config_with_branches_parses_and_lowers(fmt):
    rng = np.random.default_rng(2)
    r = lambda *s: rng.normal(0, 0.3, s).astype(np.float32)
    w = {'c1': {'kernel': r(3, 3, 1, 8), 'bias': r(8)}, 'c2': {'kernel': r(3, 3, 8, 8), 'bias': r(8)},
         'bn': {'gamma': 1 + r(8), 'beta': r(8), 'moving_mean': r(8), 'moving_variance': 1 + np.abs(r(8))},
         'sep': {'depthwise_kernel': r(3, 3, 8, 1), 'pointwise_kernel': r(1, 1, 8, 8), 'bias': r(8)},
         'out': {'kernel': r(17 * 5 * 16, 3), 'bias': r(3)}}
    layers, shp = KM.layers_from_keras_config(_keras_functional_config(fmt), w)
    assert shp == (68, 21, 1) and all('inputs' in L for L in layers)
    byname = {L['name']: L for L in layers}
    assert byname['add']['inputs'] == ['c1', 'bn'] and byname['cat']['inputs'] == ['add', 'sep/pointwise']
    assert byname['c2']['inputs'] == ['c1'] and byname['c2']['pad'] == (1, 1, 1, 1) and 'zp' not in byname
    x = rng.normal(0, 1, (2,) + shp).astype(np.float32)
    got = prog_interp.run(KM.compile_layers(layers, shp), x)
    assert np.abs(got - ocnn.forward(layers, x)).max() < 2e-5
    # a functional model that IS a chain still takes the sequential lowering (no 'inputs', the same program as before)
    cfg = _keras_functional_config(fmt)
    cfg['config']['layers'] = [cfg['config']['layers'][k] for k in (0, 1, 8, 10)] + [dict(cfg['config']['layers'][11])]
    for a, b in ((1, 'in'), (2, 'c1'), (3, 'mp'), (4, 'fl')):
        cfg['config']['layers'][a] = dict(cfg['config']['layers'][a], inbound_nodes=[[[b, 0, 0, {}]]])
    w['out'] = {'kernel': r(17 * 5 * 8, 3), 'bias': r(3)}
    chain, _ = KM.layers_from_keras_config(cfg, w)
    assert not any('inputs' in L for L in chain)
    # shared layers and several outputs are refused with a message, not mis-computed
    bad = _keras_functional_config(fmt)
    bad['config']['layers'][1]['inbound_nodes'] = bad['config']['layers'][1]['inbound_nodes'] * 2
    with pytest.raises(NotImplementedError, match='called 2 times'):
        KM.layers_from_keras_config(bad, w)


def test_functional_graph_model_file_round_trip(tmp_path):
    """A branched functional model through the file path `Segmenter` takes (keras_model.load_model_file, .npz form written by
    tools/convert_keras_hdf5.py: 'model_config' JSON + '<layer>/<weight>' arrays): parsed, lowered and executed like the oracle."""
    import json
    rng = np.random.default_rng(4)
    r = lambda *s: rng.normal(0, 0.3, s).astype(np.float32)
    w = {'c1': {'kernel': r(3, 3, 1, 8), 'bias': r(8)}, 'c2': {'kernel': r(3, 3, 8, 8), 'bias': r(8)},
         'bn': {'gamma': 1 + r(8), 'beta': r(8), 'moving_mean': r(8), 'moving_variance': 1 + np.abs(r(8))},
         'sep': {'depthwise_kernel': r(3, 3, 8, 1), 'pointwise_kernel': r(1, 1, 8, 8), 'bias': r(8)},
         'out': {'kernel': r(17 * 5 * 16, 3), 'bias': r(3)}}
    arrays = {f'{ln}/{ln}/{wn}:0': a for ln, ws in w.items() for wn, a in ws.items()}
    path = str(tmp_path / 'graph_model.npz')
    np.savez(path, model_config=json.dumps(_keras_functional_config(3)), **arrays)
    layers, shp = KM.load_model_file(path)
    assert shp == (68, 21, 1) and any(L['type'] == 'add' for L in layers) and any(L['type'] == 'concatenate' for L in layers)
    x = rng.normal(0, 1, (2,) + shp).astype(np.float32)
    ref_layers, _ = KM.layers_from_keras_config(_keras_functional_config(3), w)
    got = prog_interp.run(KM.compile_layers(layers, shp), x)
    assert np.abs(got - ocnn.forward(ref_layers, x)).max() < 2e-5
