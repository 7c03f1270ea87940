"""Graph-shaped small CNNs (functional Keras models that are not a chain) for the lowering / parity tests.

`keras.models.load_model` (segmenter.py:129-131) takes any model_config; the release assets' topology is unknown, so the op
program must compute whatever a functional model expresses -- residual adds, inception-style concatenations, permutes, reshapes,
several readers of one tensor -- if only through the slow generic rows (ISS_OP_ELT).  Every net maps (68, nmel, 1) -> softmax.
Layer dicts carry 'name' and 'inputs' (oracle/keras_cnn.py vocabulary); `NETS[name](nmel, ncls, seed)` -> (layers, shape).
`random_graph(seed, nmel, ncls)` draws one from a small grammar (the fuzz test's extension to graphs).
"""
import numpy as np

IN = '__input__'


class _G:
    def __init__(self, seed, nmel):
        self.rng = np.random.default_rng(seed)
        self.L = []
        self.shape = {IN: (68, nmel, 1)}
        self.k = 0

    def _name(self, ty):
        self.k += 1
        return f'{ty}_{self.k}'

    def _add(self, d, src, shape):
        d['name'] = self._name(d['type'])
        d['inputs'] = list(src)
        self.L.append(d)
        self.shape[d['name']] = tuple(int(v) for v in shape)
        return d['name']

    @staticmethod
    def _o(size, k, s, padding):
        return -(-size // s) if padding == 'same' else (size - k) // s + 1

    def conv(self, src, kh, kw, cout, padding='valid', s=1, act='linear', bias=True):
        h, w, c = self.shape[src]
        W = self.rng.normal(0, np.sqrt(2.0 / (kh * kw * c)), (kh, kw, c, cout)).astype(np.float32)
        d = dict(type='conv2d', W=W, b=self.rng.normal(0, 0.05, cout).astype(np.float32) if bias else None, strides=(s, s),
                 padding=padding, activation=act)
        return self._add(d, [src], (self._o(h, kh, s, padding), self._o(w, kw, s, padding), cout))

    def bn(self, src):
        c = self.shape[src][2]
        r = self.rng
        d = dict(type='batchnorm', gamma=r.uniform(0.8, 1.2, c).astype(np.float32), beta=r.normal(0, 0.1, c).astype(np.float32),
                 mean=r.normal(0, 0.1, c).astype(np.float32), var=r.uniform(0.5, 1.5, c).astype(np.float32), eps=1e-3)
        return self._add(d, [src], self.shape[src])

    def act(self, src, fn='relu', **kw):
        return self._add(dict(type='activation', fn=fn, **kw), [src], self.shape[src])

    def pool(self, src, ph, pw, sh=None, sw=None, padding='valid', kind='maxpool'):
        h, w, c = self.shape[src]
        sh, sw = sh or ph, sw or pw
        d = dict(type=kind, pool=(ph, pw), strides=(sh, sw), padding=padding)
        return self._add(d, [src], (self._o(h, ph, sh, padding), self._o(w, pw, sw, padding), c))

    def gap(self, src, kind='globalavgpool'):
        return self._add(dict(type=kind), [src], (1, 1, self.shape[src][2]))

    def drop(self, src):
        return self._add(dict(type='dropout'), [src], self.shape[src])

    def flatten(self, src):
        h, w, c = self.shape[src]
        return self._add(dict(type='flatten'), [src], (1, 1, h * w * c))

    def dense(self, src, n, act='linear'):
        i = int(np.prod(self.shape[src]))
        d = dict(type='dense', W=self.rng.normal(0, np.sqrt(2.0 / i), (i, n)).astype(np.float32),
                 b=self.rng.normal(0, 0.05, n).astype(np.float32), activation=act)
        return self._add(d, [src], (1, 1, n))

    def merge(self, ty, srcs):
        return self._add(dict(type=ty), srcs, self.shape[srcs[0]])

    def concat(self, srcs, axis=-1):
        sh = [self.shape[s] for s in srcs]
        flat = all(s[0] * s[1] == 1 for s in sh)
        nd = 1 if flat else 3
        ax = axis + nd + 1 if axis < 0 else axis
        d3 = ax - 1 + (3 - nd)
        out = tuple(sum(s[d3] for s in sh) if q == d3 else sh[0][q] for q in range(3))
        return self._add(dict(type='concatenate', axis=axis), srcs, out)

    def permute(self, src, perm):
        sh = self.shape[src]
        return self._add(dict(type='permute', perm=tuple(perm)), [src], tuple(sh[p] for p in perm))

    def reshape(self, src, target):
        t = tuple(int(v) for v in target)
        return self._add(dict(type='reshape', target=t), [src], (1, 1, t[0]) if len(t) == 1 else t)

    def head(self, src, ncls):
        if self.shape[src][0] * self.shape[src][1] > 1:
            src = self.flatten(src)
        self.dense(src, ncls, 'softmax')
        return self.L, self.shape[IN]


def residual(nmel, ncls, seed):
    """conv - [conv - BN] + skip - relu - pool, twice; the second block's skip is a strided 1x1 projection."""
    g = _G(seed, nmel)
    x = g.conv(IN, 3, 3, 32, 'same', act='relu')
    y = g.bn(g.conv(x, 3, 3, 32, 'same'))
    x = g.pool(g.act(g.merge('add', [x, y])), 2, 2)
    y = g.bn(g.conv(g.act(g.bn(g.conv(x, 3, 3, 64, 'same', s=2))), 3, 3, 64, 'same'))
    p = g.conv(x, 1, 1, 64, 'valid', s=2, bias=False)
    x = g.act(g.merge('add', [y, p]))
    x = g.dense(g.flatten(g.pool(x, 2, 1)), 64, 'relu')
    return g.head(x, ncls)


def inception(nmel, ncls, seed):
    """stem - three branches (1x1, 3x3 'same', 3x3 max-pool + 1x1) - channel concatenation (56 channels: padded to 64 for the conv
    behind it) - conv - global average pool."""
    g = _G(seed, nmel)
    x = g.pool(g.conv(IN, 3, 3, 16, act='relu'), 2, 2)
    a = g.conv(x, 1, 1, 24, act='relu')
    b = g.act(g.bn(g.conv(x, 3, 3, 24, 'same')))
    c = g.conv(g.pool(x, 3, 3, 1, 1, 'same', kind='avgpool' if nmel == 24 else 'maxpool'), 1, 1, 8, act='relu')   # (a padded mean)
    x = g.concat([a, b, c])
    x = g.conv(x, 3, 3, 64, act='relu')
    return g.head(g.gap(x), ncls)


def permute_reshape(nmel, ncls, seed):
    """Permute((2, 1, 3)) between convolutions, a Reshape to a vector and back to a map, Permute((3, 1, 2)) in front of the head."""
    g = _G(seed, nmel)
    x = g.pool(g.conv(IN, 3, 3, 8, act='relu'), 2, 2)                  # (33, (nmel - 2) // 2, 8)
    x = g.permute(x, (1, 0, 2))
    x = g.conv(x, 3, 3, 16, act='relu')
    h, w, c = g.shape[x]
    x = g.reshape(g.reshape(x, (h * w * c,)), (w, h, c))               # a different map over the same floats
    x = g.act(g.conv(x, 1, 1, 32), 'tanh')
    x = g.permute(x, (2, 0, 1))
    return g.head(g.pool(x, 2, 2), ncls)


def merges(nmel, ncls, seed):
    """The network input read by three convolutions; Maximum, Average of three, Subtract, Multiply, Minimum; concatenations along
    the H and W axes; a tensor with four readers."""
    g = _G(seed, nmel)
    a = g.conv(IN, 5, 5, 16, 'same', act='relu')
    b = g.conv(IN, 3, 3, 16, 'same', act='tanh')
    c = g.conv(IN, 1, 1, 16, 'same', act='sigmoid')
    m = g.merge('maximum', [a, b])
    v = g.merge('average', [a, b, c])
    s = g.merge('subtract', [m, v])
    p = g.merge('multiply', [s, c])
    q = g.merge('minimum', [p, a])
    x = g.pool(g.concat([q, g.merge('average', [m, v])], axis=1), 4, 3)       # along H: (136, nmel, 16) -> pooled
    y = g.pool(g.concat([q, s], axis=2), 8, 6)                                 # along W
    x = g.gap(g.conv(x, 3, 3, 32, act='relu'))
    y = g.gap(g.conv(y, 3, 3, 32, act='relu'), 'globalmaxpool')
    return g.head(g.concat([x, y]), ncls)


def dense_skip(nmel, ncls, seed):
    """A head with a skip connection and a concatenation of feature vectors behind a plain conv trunk."""
    g = _G(seed, nmel)
    x = g.pool(g.conv(IN, 4, 5, 32, act='relu'), 2, 2)
    x = g.pool(g.conv(x, 3, 3, 32, act='relu'), 2, 2)
    f = g.flatten(x)
    d1 = g.dense(f, 48, 'relu')
    d2 = g.dense(g.drop(d1), 48)
    d3 = g.act(g.merge('add', [d1, d2]), 'elu', alpha=0.7)
    z = g.concat([d3, g.dense(f, 20, 'tanh'), d1])
    return g.head(z, ncls)


def standin_residual(nmel, ncls, seed):
    """The stand-in's trunk (conv 4x5 - conv 5x3 - pool - 3x3 - 3x3 - pool) with a residual 3x3 'same' block between the pools:
    the fast kernels in front of and behind generic merge rows."""
    g = _G(seed, nmel)
    x = g.act(g.bn(g.conv(IN, 4, 5, 64)))
    x = g.pool(g.act(g.bn(g.conv(x, 5, 3, 64))), 2, 2)
    y = g.bn(g.conv(g.act(g.bn(g.conv(x, 3, 3, 64, 'same'))), 3, 3, 64, 'same'))
    x = g.act(g.merge('add', [x, y]))
    x = g.act(g.bn(g.conv(x, 3, 3, 128)))
    x = g.pool(g.act(g.bn(g.conv(x, 3, 3, 128))), 2, 1)
    x = g.dense(g.flatten(x), 128, 'relu')
    return g.head(x, ncls)


NETS = dict(residual=residual, inception=inception, permute_reshape=permute_reshape, merges=merges, dense_skip=dense_skip,
            standin_residual=standin_residual)


def random_graph(seed, nmel, ncls):
    """A random DAG: a stem, then 2-4 stages; each stage draws a block -- residual (1-2 'same' convs + Add / Maximum / Average with
    the skip), branches joined by Concatenate (2-3 branches of 1x1 / 3x3 / pooled 1x1), a plain conv, or a Permute -- followed by an
    optional pool; then a flatten / global-pool head with an optional vector skip."""
    g = _G(seed, nmel)
    r = np.random.default_rng(seed + 7919)
    ch = int(r.choice([8, 16, 24, 32, 48]))
    kh, kw = int(r.integers(2, 6)), int(r.integers(2, 6))
    x = g.conv(IN, kh, kw, ch, str(r.choice(['valid', 'same'])), act=str(r.choice(['relu', 'linear', 'tanh'])))
    if r.random() < 0.5:
        x = g.pool(x, 2, 2)
    for _ in range(int(r.integers(2, 5))):
        h, w, c = g.shape[x]
        kind = r.choice(['res', 'branch', 'conv', 'permute']) if min(h, w) >= 6 else 'res'
        if kind == 'res':
            y = x
            for q in range(int(r.integers(1, 3))):
                y = g.conv(y, 3, 3, c, 'same', act='linear')
                if r.random() < 0.5:
                    y = g.bn(y)
                if q == 0 and r.random() < 0.7:
                    y = g.act(y, str(r.choice(['relu', 'leaky_relu', 'sigmoid'])))
            how = str(r.choice(['add', 'add', 'maximum', 'average', 'multiply', 'subtract']))
            if how == 'multiply':                     # a gate, as real nets multiply: x * sigmoid(f(x)) (x * f(x) squares the logits' scale)
                y = g.act(y, 'sigmoid')
            x = g.merge(how, [x, y])
            if r.random() < 0.7:
                x = g.act(x)
        elif kind == 'branch':
            outs = []
            for _b in range(int(r.integers(2, 4))):
                co = int(r.choice([8, 12, 16, 24, 32]))
                t = r.choice(['1x1', '3x3', 'pool'])
                if t == '1x1':
                    outs.append(g.conv(x, 1, 1, co, act='relu'))
                elif t == '3x3':
                    outs.append(g.act(g.bn(g.conv(x, 3, 3, co, 'same'))))
                else:
                    outs.append(g.conv(g.pool(x, 3, 3, 1, 1, 'same', kind=str(r.choice(['maxpool', 'avgpool']))), 1, 1, co, act='relu'))
            x = g.concat(outs)
        elif kind == 'conv':
            co = int(r.choice([16, 32, 64]))
            x = g.conv(x, 3, 3, co, str(r.choice(['valid', 'same'])), act='relu')
        else:
            x = g.permute(x, (1, 0, 2))
        h, w, c = g.shape[x]
        if min(h, w) >= 8 and r.random() < 0.6:
            x = g.pool(x, 2, 2, kind=str(r.choice(['maxpool', 'avgpool'])))
    h, w, c = g.shape[x]
    while h * w * c > 6000 and min(h, w) >= 2:                           # keep the dense head small
        x = g.pool(x, 2, 2)
        h, w, c = g.shape[x]
    f = g.gap(x) if r.random() < 0.3 else g.flatten(x)
    d = g.dense(f, int(r.choice([32, 48, 64])), 'relu')
    if r.random() < 0.5:
        d = g.merge('add', [d, g.dense(g.drop(d), g.shape[d][2])])
    if r.random() < 0.3:
        d = g.concat([d, g.dense(f, 16, 'tanh')])
    return g.head(d, ncls)
