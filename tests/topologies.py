"""Seeded small-CNN topologies for the parity + throughput sweep (tests/test_gpu_topologies.py, tests/topology_sweep.py).

The reference's three Keras CNNs are un-vendored release assets (remote_utils.py:4-15); only their I/O contract is
visible ((68, nmel, 1) -> softmax over the labels, segmenter.py:146-163,182-204; ~1.25 M parameters each,
Dockerfile:18).  `standin` is the topology bench.py uses; every other entry moves ONE plausible design choice of such a
net off the path the kernels were tuned on (padding, filter size, channel counts, BatchNorm placement, head size,
pooling), so that neither parity nor throughput silently depends on the stand-in.
"""
import numpy as np


def _conv(rng, kh, kw, cin, cout, padding='valid', strides=(1, 1), act='linear'):
    return dict(type='conv2d', W=rng.normal(0, np.sqrt(2.0 / (kh * kw * cin)), (kh, kw, cin, cout)).astype(np.float32),
                b=rng.normal(0, 0.05, cout).astype(np.float32), strides=strides, padding=padding, activation=act)


def _bn(rng, c):
    return dict(type='batchnorm', gamma=rng.uniform(0.8, 1.2, c).astype(np.float32), beta=rng.normal(0, 0.1, c).astype(np.float32),
                mean=rng.normal(0, 0.1, c).astype(np.float32), var=rng.uniform(0.5, 1.5, c).astype(np.float32), eps=1e-3)


def _dense(rng, i, o, act):
    return dict(type='dense', W=rng.normal(0, np.sqrt(2.0 / i), (i, o)).astype(np.float32),
                b=rng.normal(0, 0.05, o).astype(np.float32), activation=act)


RELU = dict(type='activation', fn='relu')


def _out(size, k, s, padding):
    return -(-size // s) if padding == 'same' else (size - k) // s + 1


def build(spec, nmel, ncls, seed):
    """spec: list of ('conv', kh, kw, cout[, padding[, stride]]) | ('bn',) | ('relu',) | ('relu_bn',) | ('bn_relu',) |
    ('maxpool'|'avgpool', ph, pw[, sh, sw[, padding]]) | ('gap',) | ('flatten',) | ('dense', n[, act]) | ('drop',);
    the classifier `dense(ncls, softmax)` is appended."""
    rng = np.random.default_rng(seed)
    h, w, c = 68, nmel, 1
    L = []
    for item in spec:
        t = item[0]
        if t == 'conv':
            kh, kw, cout = item[1:4]
            padding = item[4] if len(item) > 4 else 'valid'
            s = item[5] if len(item) > 5 else 1
            L.append(_conv(rng, kh, kw, c, cout, padding, (s, s)))
            h, w, c = _out(h, kh, s, padding), _out(w, kw, s, padding), cout
        elif t == 'bn':
            L.append(_bn(rng, c))
        elif t == 'relu':
            L.append(dict(RELU))
        elif t in ('elu', 'leaky_relu', 'selu', 'softplus'):             # ('elu', alpha) / ('leaky_relu', alpha) / ('selu',) / ('softplus',)
            L.append(dict(type='activation', fn=t, **({'alpha': float(item[1])} if len(item) > 1 else {})))
        elif t == 'bn':
            L.append(_bn(rng, c))
        elif t == 'bn_relu':
            L += [_bn(rng, c), dict(RELU)]
        elif t == 'relu_bn':
            L += [dict(RELU), _bn(rng, c)]
        elif t in ('maxpool', 'avgpool'):
            ph, pw = item[1:3]
            sh, sw = item[3:5] if len(item) > 4 else (ph, pw)
            padding = item[5] if len(item) > 5 else 'valid'
            L.append(dict(type=t, pool=(ph, pw), strides=(sh, sw), padding=padding))
            h, w = _out(h, ph, sh, padding), _out(w, pw, sw, padding)
        elif t == 'gap':
            L.append(dict(type='globalavgpool'))
            h, w = 1, 1
        elif t == 'flatten':
            L.append(dict(type='flatten'))
            h, w, c = 1, 1, h * w * c
        elif t == 'dense':
            act = item[2] if len(item) > 2 else 'relu'
            L.append(_dense(rng, c, item[1], act))
            c = item[1]
        elif t == 'drop':
            L.append(dict(type='dropout'))
        else:
            raise ValueError(t)
    if h * w != 1:
        L.append(dict(type='flatten'))
        c = h * w * c
    L.append(_dense(rng, c, ncls, 'softmax'))
    return L, (68, nmel, 1)


HEAD = [('flatten',), ('dense', 192, 'linear'), ('bn_relu',), ('drop',), ('dense', 128)]
SPECS = {
    # bench.py's stand-in (keras_model.synthetic_ina_like draws the same shapes)
    'standin': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_same': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64, 'same'), ('bn_relu',), ('maxpool', 2, 2),
                   ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_3x3': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 3, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                  ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'vgg_same_3x3': [('conv', 3, 3, 64, 'same'), ('bn_relu',), ('conv', 3, 3, 64, 'same'), ('bn_relu',), ('maxpool', 2, 2),
                     ('conv', 3, 3, 128, 'same'), ('bn_relu',), ('conv', 3, 3, 128, 'same'), ('bn_relu',), ('maxpool', 2, 2)] + HEAD,
    'ch32_64': [('conv', 4, 5, 32), ('bn_relu',), ('conv', 5, 3, 32), ('bn_relu',), ('maxpool', 2, 2),
                ('conv', 3, 3, 64), ('bn_relu',), ('conv', 3, 3, 64), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'ch48_96': [('conv', 4, 5, 48), ('bn_relu',), ('conv', 5, 3, 48), ('bn_relu',), ('maxpool', 2, 2),
                ('conv', 3, 3, 96), ('bn_relu',), ('conv', 3, 3, 96), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'relu_then_bn': [('conv', 4, 5, 64), ('relu_bn',), ('conv', 5, 3, 64), ('relu_bn',), ('maxpool', 2, 2),
                     ('conv', 3, 3, 128), ('relu_bn',), ('conv', 3, 3, 128), ('relu_bn',), ('maxpool', 2, 1)] + HEAD,
    'dense512x4': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                   ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 2),
                   ('flatten',), ('dense', 512), ('drop',), ('dense', 512), ('drop',), ('dense', 512), ('drop',), ('dense', 512)],
    # the first dense layer wider than conv_dhl_kernel's 192-column tile (column tiles in blockIdx.y), behind a (2, 1) pool
    'dense512_first': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                       ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1),
                       ('flatten',), ('dense', 512, 'linear'), ('bn_relu',), ('drop',), ('dense', 128)],
    'conv1_same': [('conv', 4, 5, 64, 'same'), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                   ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv1_pool': [('conv', 4, 5, 64), ('bn_relu',), ('maxpool', 2, 2), ('conv', 5, 3, 64), ('bn_relu',),
                   ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128, 'same'), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_stride2': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64, 'valid', 2), ('bn_relu',),
                      ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_7x7': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 7, 7, 64), ('bn_relu',), ('maxpool', 2, 2),
                  ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'gap_head': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 2, 2),
                 ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('gap',), ('dense', 128)],
    # ---- round 5: every instantiation of the ring form (> 16 taps) and of the shared zero-padded first layer (FS) at least once
    'conv1_same_conv2_same': [('conv', 4, 5, 64, 'same'), ('bn_relu',), ('conv', 5, 3, 64, 'same'), ('bn_relu',), ('maxpool', 2, 2),
                              ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv1_same3x3_avg': [('conv', 3, 3, 64, 'same'), ('bn_relu',), ('conv', 3, 3, 64), ('bn_relu',), ('avgpool', 2, 2),
                          ('conv', 3, 3, 128, 'same'), ('bn_relu',), ('maxpool', 2, 2)] + HEAD,
    'conv1_same_nopool': [('conv', 4, 5, 64, 'same'), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',),
                          ('conv', 3, 3, 128, 'valid', 2), ('bn_relu',), ('maxpool', 2, 2)] + HEAD,
    'conv2_7x7_same_avg': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 7, 7, 64, 'same'), ('bn_relu',), ('avgpool', 2, 2),
                           ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_5x5': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 5, 64), ('bn_relu',), ('maxpool', 2, 2),
                  ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv2_4x5_same': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 4, 5, 64, 'same'), ('bn_relu',), ('maxpool', 2, 2),
                       ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'conv1_same_conv2_stride2': [('conv', 4, 5, 64, 'same'), ('bn_relu',), ('conv', 5, 3, 64, 'valid', 2), ('bn_relu',),
                                 ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('maxpool', 2, 1)] + HEAD,
    'overlap_pool_avg': [('conv', 4, 5, 64), ('bn_relu',), ('conv', 5, 3, 64), ('bn_relu',), ('maxpool', 3, 3, 2, 2),
                         ('conv', 3, 3, 128), ('bn_relu',), ('conv', 3, 3, 128), ('bn_relu',), ('avgpool', 2, 2)] + HEAD,
}


def nets(name, seed=0):
    """-> {'vad': (layers, in_shape), 'gender': (layers, in_shape)} for topology `name` (smn: 21 mel / 3 classes,
    gender: 24 mel / 2 classes, segmenter.py:190-204)."""
    return {'vad': build(SPECS[name], 21, 3, seed * 2 + 1), 'gender': build(SPECS[name], 24, 2, seed * 2 + 2)}
