"""Oracle: Keras-semantics forward pass of the small CNNs.  Test infrastructure only.

** parity unpinned ** -- the real forward lives in TensorFlow/Keras
(segmenter.py:131 `keras.models.load_model`, :163 `self.nn.predict`) over
un-vendored HDF5 assets (remote_utils.py:4-15); neither exists in this
container.  This file restates the published Keras layer semantics:
channels-last NHWC activations, HWIO conv kernels, 'valid'/'same' padding with
the TF rule (extra pad goes bottom/right), BatchNormalization inference form
gamma*(x-mean)/sqrt(var+eps)+beta, Flatten in (H,W,C) order, Dropout=identity,
softmax over the last axis.  Two independent implementations are kept so they
can check each other: `forward` (torch-CPU functional, fast enough to be the
CPU baseline) and `forward_naive` (pure numpy loops, tiny cases only).

A model is a list of plain dicts (same vocabulary as
inaspeechsegmenter_amd/keras_model.py produces, but this file does not import it):
  conv2d   : W (kh,kw,cin,cout) f32, b (cout,)|None, strides, padding, activation
  dense    : W (in,out) f32, b (out,)|None, activation
  batchnorm: gamma, beta, mean, var, eps
  activation: fn ; maxpool/avgpool: pool, strides, padding ; flatten ; dropout
  globalavgpool / globalmaxpool
  reshape  : target (keras.layers.Reshape target_shape: (n,) or (a, b, c)) ; permute: perm (0-based over (H, W, C))
Graph-shaped models (functional Keras models that are not a chain): every dict carries 'name' and 'inputs' (names of the layers it
reads, '__input__' = the network input), in any topological order, plus the merge layers
  add / subtract / multiply / average / maximum / minimum (keras.layers.Add ...: elementwise over inputs of one shape)
  concatenate : axis (Keras numbering, batch axis included; default -1)
"""
import numpy as np


def same_pads(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


_ALPHA = {'elu': 1.0, 'leaky_relu': 0.3}          # keras.activations.elu / keras.layers.LeakyReLU defaults


def _act_np(x, fn, alpha=None):
    alpha = _ALPHA.get(fn, 0.0) if alpha is None else alpha
    if fn in (None, 'linear'):
        return x
    if fn == 'relu':
        return np.maximum(x, 0)
    if fn == 'relu_general':                       # keras.layers.ReLU(max_value, negative_slope, threshold); alpha = (slope, max, thr)
        sl, mv, th = (np.float32(v) for v in alpha)
        return np.where(x > th, np.minimum(x, mv), sl * (x - th)).astype(np.float32)
    if fn == 'relu_max':                           # keras.layers.ReLU(max_value=alpha): min(max(x, 0), alpha)
        return np.minimum(np.maximum(x, 0), np.float32(alpha))
    if fn == 'elu':                                # keras.activations.elu: x if x > 0 else alpha * (exp(x) - 1)
        return np.where(x > 0, x, np.float32(alpha) * (np.exp(np.minimum(x, 0)) - 1))
    if fn == 'leaky_relu':                         # keras.layers.LeakyReLU: x if x > 0 else alpha * x
        return np.where(x > 0, x, np.float32(alpha) * x)
    if fn == 'selu':                               # scale * elu(x, alpha) with the fixed SELU constants
        return np.float32(1.05070098) * np.where(x > 0, x, np.float32(1.67326324) * (np.exp(np.minimum(x, 0)) - 1))
    if fn == 'softplus':
        return np.logaddexp(x, 0)
    if fn == 'sigmoid':
        return 1.0 / (1.0 + np.exp(-x))
    if fn == 'tanh':
        return np.tanh(x)
    if fn == 'softmax':
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e / e.sum(axis=-1, keepdims=True)
    raise ValueError(fn)


def forward(layers, x, batch_size=1024, threads=None):
    """x: (N,H,W,C) float32 -> (N,classes) float32, torch-CPU."""
    import torch
    import torch.nn.functional as F
    if threads:
        torch.set_num_threads(threads)

    def act(t, fn, alpha=None):
        alpha = _ALPHA.get(fn, 0.0) if alpha is None else alpha
        if fn in (None, 'linear'):
            return t
        if fn == 'relu':
            return torch.relu(t)
        if fn == 'relu_general':
            sl, mv, th = (float(v) for v in alpha)
            return torch.where(t > th, torch.clamp(t, max=mv), sl * (t - th))
        if fn == 'relu_max':
            return torch.clamp(t, 0.0, float(alpha))
        if fn == 'elu':
            return F.elu(t, alpha=float(alpha))
        if fn == 'leaky_relu':
            return F.leaky_relu(t, negative_slope=float(alpha))
        if fn == 'selu':
            return F.selu(t)
        if fn == 'softplus':
            return F.softplus(t)
        if fn == 'sigmoid':
            return torch.sigmoid(t)
        if fn == 'tanh':
            return torch.tanh(t)
        if fn == 'softmax':
            return torch.softmax(t, dim=-1)
        raise ValueError(fn)

    def step(L, t, flat):
        ty = L['type']
        if ty in ('conv2d', 'depthwise'):
            # Conv2D: kernel (kh, kw, cin, cout).  DepthwiseConv2D: kernel (kh, kw, cin, depth_multiplier), output channel
            # c * multiplier + m = filter m of input channel c (torch groups = cin has the same channel order).
            # dilation_rate: taps (dy, dx) apart; 'same' pads for the EFFECTIVE kernel size (k - 1) * d + 1
            if ty == 'depthwise':
                kh, kw, cin, mult = L['W'].shape
                w = torch.from_numpy(np.ascontiguousarray(L['W'].transpose(2, 3, 0, 1).reshape(cin * mult, 1, kh, kw)))
                groups = cin
            else:
                w = torch.from_numpy(np.ascontiguousarray(L['W'].transpose(3, 2, 0, 1)))
                groups = 1
            b = None if L.get('b') is None else torch.from_numpy(L['b'])
            kh, kw = L['W'].shape[:2]
            sh, sw = L.get('strides', (1, 1))
            dy, dx = L.get('dilation', (1, 1))
            if L.get('pad'):                       # a ZeroPadding2D in front of the convolution: (top, bottom, left, right)
                zt, zb, zl, zr = L['pad']
                t = F.pad(t, (zl, zr, zt, zb))
            if L.get('padding', 'valid') == 'same':
                pt, pb = same_pads(t.shape[2], (kh - 1) * dy + 1, sh)
                pl, pr = same_pads(t.shape[3], (kw - 1) * dx + 1, sw)
                t = F.pad(t, (pl, pr, pt, pb))
            t = act_nchw(F.conv2d(t, w, b, stride=(sh, sw), dilation=(dy, dx), groups=groups), L.get('activation'), act, L.get('alpha'))
        elif ty == 'batchnorm':
            sc = L['gamma'] / np.sqrt(L['var'] + np.float32(L['eps']))
            sh_ = L['beta'] - L['mean'] * sc
            sc_t = torch.from_numpy(sc.astype(np.float32))
            sh_t = torch.from_numpy(sh_.astype(np.float32))
            if flat:
                t = t * sc_t + sh_t
            else:
                t = t * sc_t[None, :, None, None] + sh_t[None, :, None, None]
        elif ty == 'activation':
            t = act(t, L['fn'], L.get('alpha')) if flat else act_nchw(t, L['fn'], act, L.get('alpha'))
        elif ty in ('maxpool', 'avgpool'):
            ph, pw = L['pool']
            sh, sw = L.get('strides') or L['pool']
            if L.get('padding', 'valid') == 'same':
                pt, pb = same_pads(t.shape[2], ph, sh)
                pl, pr = same_pads(t.shape[3], pw, sw)
                fill = float('-inf') if ty == 'maxpool' else 0.0
                if ty == 'avgpool' and (pt + pb + pl + pr):    # Keras / TF: the mean is over the elements inside the input
                    ones = F.pad(torch.ones_like(t[:1, :1]), (pl, pr, pt, pb))
                    cnt = F.avg_pool2d(ones, (ph, pw), (sh, sw)) * (ph * pw)
                    t = F.avg_pool2d(F.pad(t, (pl, pr, pt, pb)), (ph, pw), (sh, sw)) * (ph * pw) / cnt
                    return t, flat
                t = F.pad(t, (pl, pr, pt, pb), value=fill)
            t = F.max_pool2d(t, (ph, pw), (sh, sw)) if ty == 'maxpool' else F.avg_pool2d(t, (ph, pw), (sh, sw))
        elif ty == 'globalavgpool':
            t = t.mean(dim=(2, 3)); flat = True
        elif ty == 'globalmaxpool':
            t = t.amax(dim=(2, 3)); flat = True
        elif ty == 'flatten':
            t = t.permute(0, 2, 3, 1).reshape(t.shape[0], -1); flat = True
        elif ty == 'dense':
            assert flat, "dense on un-flattened input"
            t = t @ torch.from_numpy(L['W'])
            if L.get('b') is not None:
                t = t + torch.from_numpy(L['b'])
            t = act(t, L.get('activation'), L.get('alpha'))
        elif ty == 'dropout':
            pass
        elif ty == 'reshape':                      # row-major on the channels-last tensor
            u = (t if flat else t.permute(0, 2, 3, 1)).reshape((t.shape[0],) + tuple(int(v) for v in L['target']))
            if u.dim() == 2:
                t, flat = u, True
            elif u.dim() == 4:
                t, flat = u.permute(0, 3, 1, 2), False
            else:
                raise ValueError(f"reshape target {L['target']}")
        elif ty == 'permute':                      # output axis i = input axis perm[i] of (H, W, C)
            assert not flat
            pm = [int(v) + 1 for v in L['perm']]
            t = t.permute(0, 2, 3, 1).permute(0, *pm).permute(0, 3, 1, 2)
        else:
            raise ValueError(ty)
        return t, flat

    def merge(L, ins):
        ty = L['type']
        ts = [a for a, _ in ins]
        flat = ins[0][1]
        assert all(f == flat for _, f in ins), "merge of flat and (H, W, C) tensors"
        if ty == 'concatenate':
            nd = 1 if flat else 3
            ax = int(L.get('axis', -1))
            ax = ax + nd + 1 if ax < 0 else ax             # Keras axis, batch included: 1..nd
            dim = 1 if flat else {1: 2, 2: 3, 3: 1}[ax]    # NHWC axis -> the NCHW dim used here
            return torch.cat(ts, dim=dim), flat
        r = ts[0]
        for u in ts[1:]:
            r = {'add': torch.add, 'average': torch.add, 'subtract': torch.sub, 'multiply': torch.mul,
                 'maximum': torch.maximum, 'minimum': torch.minimum}[ty](r, u)
        return (r / np.float32(len(ts)) if ty == 'average' else r), flat

    graph = any('inputs' in L for L in layers)
    outs = []
    with torch.no_grad():
        for s in range(0, len(x), batch_size):
            t = torch.from_numpy(np.ascontiguousarray(x[s:s + batch_size], dtype=np.float32))
            t = t.permute(0, 3, 1, 2)                      # NCHW internally
            flat = False
            if not graph:
                for L in layers:
                    t, flat = step(L, t, flat)
            else:
                t, flat = _run_graph(layers, (t, flat), lambda L, v: step(L, v[0], v[1]), merge)
            outs.append(t.numpy())
    return np.concatenate(outs) if outs else np.zeros((0, 0), np.float32)


_MERGE_TYPES = ('add', 'subtract', 'multiply', 'average', 'maximum', 'minimum', 'concatenate')


def _run_graph(layers, x, step, merge):
    """Evaluate a graph-shaped layer list: every layer once all of its inputs exist; the value of the one layer nobody reads is
    the model's output."""
    vals = {'__input__': x}
    names = [L['name'] for L in layers]
    srcs = [L.get('inputs') or [names[k - 1] if k else '__input__'] for k, L in enumerate(layers)]
    read = {nm for src in srcs for nm in src}
    sinks = [nm for nm in names if nm not in read]
    assert len(sinks) == 1, sinks
    left = list(range(len(layers)))
    while left:
        k = next((k for k in left if all(nm in vals for nm in srcs[k])), None)
        if k is None:
            raise ValueError('cycle or missing input in the layer graph')
        L = layers[k]
        vals[L['name']] = merge(L, [vals[nm] for nm in srcs[k]]) if L['type'] in _MERGE_TYPES else step(L, vals[srcs[k][0]])
        left.remove(k)
    return vals[sinks[0]]


def act_nchw(t, fn, act, alpha=None):
    if fn == 'softmax':
        return act(t.permute(0, 2, 3, 1), fn).permute(0, 3, 1, 2)
    return act(t, fn, alpha)


def forward_naive(layers, x):
    """Pure-numpy NHWC loops, float32, for tiny shapes: independent check of `forward`."""
    t = np.asarray(x, dtype=np.float32)
    if any('inputs' in L for L in layers):
        def merge(L, ts):
            if L['type'] == 'concatenate':
                ax = int(L.get('axis', -1))
                return np.concatenate(ts, axis=ax if ax >= 0 else ts[0].ndim + ax)      # NHWC / (N, n): Keras' own axis numbering
            f = {'add': np.add, 'average': np.add, 'subtract': np.subtract, 'multiply': np.multiply, 'maximum': np.maximum,
                 'minimum': np.minimum}[L['type']]
            r = ts[0]
            for u in ts[1:]:
                r = f(r, u)
            return (r / np.float32(len(ts)) if L['type'] == 'average' else r).astype(np.float32)
        return _run_graph(layers, t, lambda L, v: _naive_chain([L], v), merge)
    return _naive_chain(layers, t)


def _naive_chain(layers, t):
    for L in layers:
        ty = L['type']
        if ty == 'reshape':
            t = t.reshape((t.shape[0],) + tuple(int(v) for v in L['target']))
        elif ty == 'permute':
            t = t.transpose((0,) + tuple(int(v) + 1 for v in L['perm']))
        elif ty == 'conv2d':
            W = L['W']; kh, kw, cin, cout = W.shape
            sh, sw = L.get('strides', (1, 1))
            if L.get('pad'):
                zt, zb, zl, zr = L['pad']
                t = np.pad(t, ((0, 0), (zt, zb), (zl, zr), (0, 0)))
            if L.get('padding', 'valid') == 'same':
                pt, pb = same_pads(t.shape[1], kh, sh); pl, pr = same_pads(t.shape[2], kw, sw)
                t = np.pad(t, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
            n, H, Wd, _ = t.shape
            ho, wo = (H - kh) // sh + 1, (Wd - kw) // sw + 1
            o = np.zeros((n, ho, wo, cout), np.float32)
            for y in range(ho):
                for xx in range(wo):
                    patch = t[:, y * sh:y * sh + kh, xx * sw:xx * sw + kw, :].reshape(n, -1)
                    o[:, y, xx, :] = patch @ W.reshape(-1, cout)
            if L.get('b') is not None:
                o = o + L['b']
            t = _act_np(o, L.get('activation'), L.get('alpha')).astype(np.float32)
        elif ty == 'batchnorm':
            sc = L['gamma'] / np.sqrt(L['var'] + np.float32(L['eps']))
            t = (t * sc + (L['beta'] - L['mean'] * sc)).astype(np.float32)
        elif ty == 'activation':
            t = _act_np(t, L['fn'], L.get('alpha')).astype(np.float32)
        elif ty in ('maxpool', 'avgpool'):
            ph, pw = L['pool']; sh, sw = L.get('strides') or L['pool']
            if L.get('padding', 'valid') == 'same':
                pt, pb = same_pads(t.shape[1], ph, sh); pl, pr = same_pads(t.shape[2], pw, sw)
                t = np.pad(t, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf if ty == 'maxpool' else np.nan)
            n, H, Wd, c = t.shape
            ho, wo = (H - ph) // sh + 1, (Wd - pw) // sw + 1
            o = np.zeros((n, ho, wo, c), np.float32)
            for y in range(ho):
                for xx in range(wo):
                    win = t[:, y * sh:y * sh + ph, xx * sw:xx * sw + pw, :]
                    o[:, y, xx, :] = win.max(axis=(1, 2)) if ty == 'maxpool' else np.nanmean(win, axis=(1, 2))   # (NaN = padding)
            t = o
        elif ty == 'globalavgpool':
            t = t.mean(axis=(1, 2))
        elif ty == 'globalmaxpool':
            t = t.max(axis=(1, 2))
        elif ty == 'flatten':
            t = t.reshape(t.shape[0], -1)
        elif ty == 'dense':
            t = t @ L['W']
            if L.get('b') is not None:
                t = t + L['b']
            t = _act_np(t, L.get('activation'), L.get('alpha')).astype(np.float32)
        elif ty == 'dropout':
            pass
        else:
            raise ValueError(ty)
    return t


def flops_per_sample(layers, in_shape):
    """2*MAC count of conv + dense layers for one (H,W,C) input (SURVEY 8d F_net)."""
    h, w, c = in_shape
    fl = 0
    flat = None
    for L in layers:
        ty = L['type']
        if ty == 'conv2d':
            kh, kw, cin, cout = L['W'].shape
            sh, sw = L.get('strides', (1, 1))
            if L.get('pad'):
                h, w = h + L['pad'][0] + L['pad'][1], w + L['pad'][2] + L['pad'][3]
            if L.get('padding', 'valid') == 'same':
                h, w = -(-h // sh), -(-w // sw)
            else:
                h, w = (h - kh) // sh + 1, (w - kw) // sw + 1
            fl += 2 * kh * kw * cin * cout * h * w
            c = cout
        elif ty in ('maxpool', 'avgpool'):
            ph, pw = L['pool']; sh, sw = L.get('strides') or L['pool']
            if L.get('padding', 'valid') == 'same':
                h, w = -(-h // sh), -(-w // sw)
            else:
                h, w = (h - ph) // sh + 1, (w - pw) // sw + 1
        elif ty == 'flatten':
            flat = h * w * c
        elif ty in ('globalavgpool', 'globalmaxpool'):
            flat = c
        elif ty == 'dense':
            fl += 2 * L['W'].shape[0] * L['W'].shape[1]
            flat = L['W'].shape[1]
    return fl
