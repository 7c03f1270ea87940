"""Keras-model front end of the CNN engine: HDF5 / model_config -> layer list -> op program.

The reference loads its three CNNs with `keras.models.load_model(path, compile=False)`
(segmenter.py:129-131) from release assets named in remote_utils.py:7-15; the topology is
NOT in the reference tree, so nothing here hard-codes one: `layers_from_keras_config`
interprets the `model_config` JSON stored in the HDF5 file and `compile_layers` lowers the
resulting layer list onto the op program of include/iss.h (ISS_OP_*), fusing
bias / BatchNormalization / activation into the conv epilogue.

Keras conventions honoured (Keras documentation; not visible in the reference):
channels-last activations, HWIO conv kernels, (in,out) dense kernels, TF 'same' padding
(extra pad bottom/right), BatchNormalization inference form with `epsilon`, Flatten in
(H,W,C) order, Dropout = identity.

File formats accepted by `load_model_file`:
  *.hdf5 / *.h5   Keras HDF5: h5py where it is installed, else the package's own reader of the file format (hdf5_reader.py)
  *.npz           flat export written by tools/convert_keras_hdf5.py (numpy only):
                  key 'model_config' (JSON string) + one array per '<layer>/<weight>'.
"""
import json
import os

import numpy as np

from . import _native as N

_ACT_CODE = {None: 0, 'linear': 0, 'relu': 1, 'sigmoid': 2, 'tanh': 3}                 # fused into a conv / dense epilogue
_ACT_OP_CODE = {'elu': 4, 'leaky_relu': 5, 'selu': 6, 'softplus': 7, 'relu_max': 8, 'relu_general': 9}    # their own elementwise op (ISS_OP_ACT)
_ACT_DEFAULT_ALPHA = {'elu': 1.0, 'leaky_relu': 0.3}      # keras.activations.elu / keras.layers.LeakyReLU defaults


class CompiledNet:
    def __init__(self, prog, blob, buf_elems, in_shape, out_dim, flops, patch_input):
        self.prog = prog
        self.blob = blob
        self.buf_elems = buf_elems
        self.in_shape = in_shape
        self.out_dim = out_dim
        self.flops_per_sample = flops
        self.patch_input = patch_input


# ------------------------------------------------------------------------------ Keras parsing
def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def _string_act_alpha(fn):
    """The activation STRING 'leaky_relu' (`Activation('leaky_relu')`, `Conv2D(activation='leaky_relu')`) is
    keras.activations.leaky_relu, whose negative_slope defaults to 0.2; only the LeakyReLU LAYER defaults to 0.3."""
    return {'alpha': 0.2} if fn == 'leaky_relu' else {}


def layers_from_keras_config(model_config, weights):
    """model_config: dict (parsed JSON of the HDF5 'model_config' attribute).
    weights: dict layer_name -> dict short_weight_name -> ndarray
             (short name = 'kernel','bias','gamma','beta','moving_mean','moving_variance').
    Returns (layers, input_shape(H,W,C))."""
    cls = model_config.get('class_name')
    cfg = model_config['config']
    klayers = cfg['layers'] if isinstance(cfg, dict) else cfg
    if cls not in ('Sequential', 'Model', 'Functional'):
        raise NotImplementedError(f"Keras model class {cls!r}")
    functional = cls != 'Sequential'
    inbound = {}                                    # Keras layer name -> names of the layers it reads (functional models)
    if functional:
        for kl in klayers:
            inbound[kl['config'].get('name', kl.get('name'))] = _inbound_names(kl)
        if isinstance(cfg, dict) and (len(cfg.get('input_layers', [0])) != 1 or len(_flat_refs(cfg.get('output_layers', [0]))) > 1):
            raise NotImplementedError("models with several inputs or outputs: the segmenter networks have one of each")
    layers, in_shape = [], None
    produced = {}                                   # Keras layer name -> (first, last) index into `layers` (None: an InputLayer)
    for kl in klayers:
        cn, c = kl['class_name'], kl['config']
        name = c.get('name')
        n_before = len(layers)
        produced[name] = None
        if in_shape is None and c.get('batch_input_shape') is not None:
            in_shape = tuple(int(v) for v in c['batch_input_shape'][1:])
        if in_shape is None and c.get('batch_shape') is not None:
            in_shape = tuple(int(v) for v in c['batch_shape'][1:])
        w = weights.get(name, {})
        if cn == 'InputLayer':
            continue
        if cn == 'ZeroPadding2D':                   # merged into the convolution behind it (explicit padding)
            pd = c.get('padding', 1)
            if np.isscalar(pd):
                zp = (int(pd),) * 4
            elif np.isscalar(pd[0]):
                zp = (int(pd[0]), int(pd[0]), int(pd[1]), int(pd[1]))
            else:
                zp = (int(pd[0][0]), int(pd[0][1]), int(pd[1][0]), int(pd[1][1]))
            if c.get('data_format', 'channels_last') != 'channels_last':
                raise NotImplementedError('channels_first ZeroPadding2D')
            layers.append(dict(type='zeropad', name=name, pad=zp))
            produced[name] = (n_before, n_before)
            continue
        if cn in ('Conv2D', 'Convolution2D'):
            if c.get('data_format', 'channels_last') != 'channels_last':
                raise NotImplementedError('channels_first Conv2D')
            layers.append(dict(type='conv2d', name=name, W=np.asarray(w['kernel'], np.float32),
                               b=np.asarray(w['bias'], np.float32) if c.get('use_bias', True) else None,
                               strides=_pair(c.get('strides', 1)), padding=c.get('padding', 'valid'),
                               activation=c.get('activation', 'linear'), dilation=_pair(c.get('dilation_rate', 1)),
                               **_string_act_alpha(c.get('activation'))))
        elif cn in ('DepthwiseConv2D', 'SeparableConv2D'):
            # no kernel of their own: lowered as ordinary convolutions on zero-filled kernels (expand_generic_layers) -- slow
            # for wide layers, but a model_config that uses them loads and computes what Keras computes
            if c.get('data_format', 'channels_last') != 'channels_last':
                raise NotImplementedError(f'channels_first {cn}')
            dk = np.asarray(w['depthwise_kernel'], np.float32)
            sep = cn == 'SeparableConv2D'
            layers.append(dict(type='depthwise', name=name + ('/depthwise' if sep else ''), W=dk,
                               b=None if sep or not c.get('use_bias', True) else np.asarray(w['bias'], np.float32),
                               strides=_pair(c.get('strides', 1)), padding=c.get('padding', 'valid'),
                               activation='linear' if sep else c.get('activation', 'linear'), dilation=_pair(c.get('dilation_rate', 1)),
                               **({} if sep else _string_act_alpha(c.get('activation')))))
            if sep:
                layers.append(dict(type='conv2d', name=name + '/pointwise', W=np.asarray(w['pointwise_kernel'], np.float32),
                                   b=np.asarray(w['bias'], np.float32) if c.get('use_bias', True) else None,
                                   strides=(1, 1), padding='valid', activation=c.get('activation', 'linear'), dilation=(1, 1),
                                   **_string_act_alpha(c.get('activation'))))
        elif cn == 'Dense':
            layers.append(dict(type='dense', name=name, W=np.asarray(w['kernel'], np.float32),
                               b=np.asarray(w['bias'], np.float32) if c.get('use_bias', True) else None,
                               activation=c.get('activation', 'linear'), **_string_act_alpha(c.get('activation'))))
        elif cn == 'BatchNormalization':
            axis = c.get('axis', -1)
            axis = axis[0] if isinstance(axis, (list, tuple)) else axis
            if axis not in (-1, 3, 1):
                raise NotImplementedError(f'BatchNormalization axis {axis}')
            n = len(w['moving_mean'])
            layers.append(dict(type='batchnorm', name=name,
                               gamma=np.asarray(w['gamma'], np.float32) if c.get('scale', True) else np.ones(n, np.float32),
                               beta=np.asarray(w['beta'], np.float32) if c.get('center', True) else np.zeros(n, np.float32),
                               mean=np.asarray(w['moving_mean'], np.float32),
                               var=np.asarray(w['moving_variance'], np.float32), eps=float(c.get('epsilon', 1e-3))))
        elif cn == 'Activation':
            layers.append(dict(type='activation', name=name, fn=c['activation'], **_string_act_alpha(c['activation'])))
        elif cn == 'ReLU':
            slope = float(c.get('negative_slope', 0.0) or 0.0)
            thr = float(c.get('threshold', 0.0) or 0.0)
            if thr != 0.0 or (c.get('max_value') is not None and slope != 0.0):
                # keras.layers.ReLU in full: x > threshold ? min(x, max_value) : negative_slope * (x - threshold)
                mv = float(c['max_value']) if c.get('max_value') is not None else float('inf')
                layers.append(dict(type='activation', name=name, fn='relu_general', alpha=(slope, mv, thr)))
            elif c.get('max_value') is not None:           # min(max(x, 0), max_value): its own elementwise op
                layers.append(dict(type='activation', name=name, fn='relu_max', alpha=float(c['max_value'])))
            else:
                layers.append(dict(type='activation', name=name, fn='relu') if slope == 0.0 else
                              dict(type='activation', name=name, fn='leaky_relu', alpha=slope))
        elif cn == 'LeakyReLU':
            layers.append(dict(type='activation', name=name, fn='leaky_relu',
                               alpha=float(c.get('alpha', c.get('negative_slope', 0.3)))))
        elif cn == 'ELU':
            layers.append(dict(type='activation', name=name, fn='elu', alpha=float(c.get('alpha', 1.0))))
        elif cn == 'Softmax':
            layers.append(dict(type='activation', name=name, fn='softmax'))
        elif cn in ('MaxPooling2D', 'AveragePooling2D'):
            pool = _pair(c.get('pool_size', 2))
            st = c.get('strides')
            layers.append(dict(type='maxpool' if cn.startswith('Max') else 'avgpool', name=name, pool=pool,
                               strides=_pair(st) if st is not None else pool, padding=c.get('padding', 'valid')))
        elif cn == 'GlobalAveragePooling2D':
            layers.append(dict(type='globalavgpool', name=name))
        elif cn == 'GlobalMaxPooling2D':
            layers.append(dict(type='globalmaxpool', name=name))
        elif cn == 'Flatten':
            layers.append(dict(type='flatten', name=name))
        elif cn in ('Dropout', 'SpatialDropout2D', 'GaussianNoise', 'GaussianDropout', 'AlphaDropout', 'ActivityRegularization'):
            layers.append(dict(type='dropout', name=name))
        elif cn == 'Reshape':
            layers.append(dict(type='reshape', name=name, target=tuple(int(v) for v in c['target_shape'])))
        elif cn == 'Permute':                       # dims are 1-based over the non-batch axes; stored activations are (H, W, C)
            dims = tuple(int(v) for v in c['dims'])
            if sorted(dims) == [1]:
                layers.append(dict(type='dropout', name=name))
            elif sorted(dims) == [1, 2, 3]:
                layers.append(dict(type='permute', name=name, perm=tuple(d - 1 for d in dims)))
            else:
                raise NotImplementedError(f"Permute{dims}: the op program stores (H, W, C) activations")
        elif cn in ('Add', 'Subtract', 'Multiply', 'Average', 'Maximum', 'Minimum', 'Concatenate'):
            if not functional:
                raise ValueError(f"{cn} in a Sequential model")
            layers.append(dict(type=cn.lower(), name=name, **({'axis': int(c.get('axis', -1))} if cn == 'Concatenate' else {})))
        else:
            raise NotImplementedError(f"Keras layer {cn!r} ({name}) is not supported by the op program")
        if len(layers) > n_before:
            produced[name] = (n_before, len(layers) - 1)
    if functional and not _wire_graph(layers, klayers, inbound, produced):
        functional = False                          # a plain chain in list order: the sequential lowering takes it
    # ZeroPadding2D -> explicit padding of the Conv2D right behind it (the only place the op program can express it)
    merged = []
    if functional:                                  # graph: the padding layer's one reader must be that convolution
        byname = {L['name']: L for L in layers}
        nread = {}
        for L in layers:
            for nm in L['inputs']:
                nread[nm] = nread.get(nm, 0) + 1
        for L in layers:
            if L['type'] == 'zeropad':
                continue
            src = byname.get(L['inputs'][0]) if len(L['inputs']) == 1 else None
            if src is not None and src['type'] == 'zeropad':
                if L['type'] != 'conv2d' or L.get('padding', 'valid') != 'valid' or nread[src['name']] != 1:
                    raise NotImplementedError(f"ZeroPadding2D ({src['name']}) must be read by one Conv2D(padding='valid') only")
                L = dict(L, pad=src['pad'], inputs=list(src['inputs']))
            if any(byname.get(nm, {}).get('type') == 'zeropad' for nm in L['inputs']):
                raise NotImplementedError(f"ZeroPadding2D in front of {L['name']} ({L['type']})")
            merged.append(L)
        if any(L['type'] == 'zeropad' and not nread.get(L['name']) for L in layers):
            raise NotImplementedError('ZeroPadding2D at the end of the model')
    else:
        for L in layers:
            if merged and merged[-1]['type'] == 'zeropad':
                z = merged.pop()
                if L['type'] != 'conv2d' or L.get('padding', 'valid') != 'valid':
                    raise NotImplementedError(f"ZeroPadding2D ({z['name']}) must be followed by a Conv2D(padding='valid')")
                L = dict(L, pad=z['pad'])
            merged.append(L)
        if merged and merged[-1]['type'] == 'zeropad':
            raise NotImplementedError('ZeroPadding2D at the end of the model')
    layers = merged
    if in_shape is None:
        raise ValueError("model_config carries no batch_input_shape")
    if len(in_shape) == 1:                          # plain MLP on feature vectors (e.g. the x-vector gender model)
        in_shape = (1, 1, in_shape[0])
    if len(in_shape) != 3:
        raise NotImplementedError(f"input shape {in_shape}: need (H, W, C) or (C,)")
    return layers, in_shape


def _flat_refs(o):
    """[name, node, tensor] references in a Keras config value (`input_layers` / `output_layers`: one reference or a list of them)."""
    if isinstance(o, (list, tuple)):
        if len(o) >= 3 and isinstance(o[0], str) and isinstance(o[1], int) and isinstance(o[2], int):
            return [o[0]]
        return [r for v in o for r in _flat_refs(v)]
    return []


def _inbound_names(kl):
    """Names of the layers whose outputs a functional-model layer reads.  Keras 2 writes `inbound_nodes` as
    [[[name, node, tensor, kwargs], ...]], Keras 3 as [{'args': [<__keras_tensor__ with keras_history [name, node, tensor]> | list
    of them], 'kwargs': {...}}]; a layer called more than once (a shared layer) has several nodes."""
    inb = kl.get('inbound_nodes', [])
    if not inb:
        return []
    if len(inb) > 1:
        raise NotImplementedError(f"layer {kl['config'].get('name')!r} is called {len(inb)} times (shared layers are not lowered)")
    names = []

    def walk(o):
        if isinstance(o, dict):
            if o.get('class_name') == '__keras_tensor__':
                names.append(o['config']['keras_history'][0])
                return
            for v in o.values():
                walk(v)
        elif isinstance(o, (list, tuple)):
            if len(o) >= 3 and isinstance(o[0], str) and isinstance(o[1], int) and isinstance(o[2], int):
                names.append(o[0])
                return
            for v in o:
                walk(v)
    walk(inb[0])
    return names


def _wire_graph(layers, klayers, inbound, produced):
    """Give every parsed layer of a functional model its 'inputs' (names of parsed layers, GRAPH_INPUT for the InputLayer).  Returns
    False -- and leaves the list untouched -- when the model is a plain chain in list order."""
    def out_name(kname):                            # parsed layer that carries a Keras layer's output
        if kname not in produced:
            raise ValueError(f"inbound layer {kname!r} is not in the model")
        pr = produced[kname]
        return GRAPH_INPUT if pr is None else layers[pr[1]]['name']
    wired, chain, prev = [], True, GRAPH_INPUT
    for kl in klayers:
        kname = kl['config'].get('name')
        pr = produced[kname]
        if pr is None:
            continue
        src = [out_name(nm) for nm in inbound.get(kname, [])]
        if not src:
            raise ValueError(f"layer {kname!r} of a functional model has no inbound node")
        for q in range(pr[0], pr[1] + 1):
            wired.append(src if q == pr[0] else [layers[q - 1]['name']])
            chain = chain and wired[-1] == [prev]
            prev = layers[q]['name']
    if chain:
        return False
    for L, src in zip(layers, wired):
        L['inputs'] = src
    return True


def _short(wname):
    s = wname.split('/')[-1]
    return s.split(':')[0]


def load_model_file(path):
    """-> (layers, input_shape).  See module docstring for formats."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        z = np.load(path, allow_pickle=False)
        cfg = json.loads(str(z['model_config']))
        weights = {}
        for k in z.files:
            if k == 'model_config':
                continue
            lname, wname = k.rsplit('/', 1) if '/' in k else (k, k)
            weights.setdefault(lname.split('/')[0], {})[_short(wname)] = z[k]
        return layers_from_keras_config(cfg, weights)
    try:                                            # h5py where it exists; else the package's own reader of the file format
        import h5py
        opener = lambda p: h5py.File(p, 'r')        # noqa: E731
    except ImportError:
        from . import hdf5_reader
        opener = hdf5_reader.File
    with opener(path) as f:
        mc = f.attrs['model_config']
        if isinstance(mc, bytes):
            mc = mc.decode('utf-8')
        cfg = json.loads(mc)
        g = f['model_weights'] if 'model_weights' in f else f
        weights = {}
        for lname in g:
            names = g[lname].attrs.get('weight_names', [])
            for wn in names:
                wn = wn.decode('utf-8') if isinstance(wn, bytes) else wn
                weights.setdefault(lname, {})[_short(wn)] = np.asarray(g[lname][wn])
    return layers_from_keras_config(cfg, weights)


# ------------------------------------------------------------------------------ lowering
def _same_pads(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


class _Builder:
    def __init__(self):
        self.rows = []
        self.blob = []
        self.nblob = 0
        self.buf_elems = {}
        self.flops = 0

    def add_blob(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        off = self.nblob
        pad = (-a.size) % 8                     # 8 floats: keeps the bf16 hi/lo copies 16-byte aligned
        self.blob.append(a)
        if pad:
            self.blob.append(np.zeros(pad, np.float32))
        self.nblob += a.size + pad
        return off

    def use_buf(self, b, elems):
        self.buf_elems[b] = max(self.buf_elems.get(b, 0), int(elems))

    def conv(self, src, dst, shape_in, Wm, kh, kw, sh, sw, pt, pl, ho, wo, bias=None, act=0, ps=None, pt_=None,
             res=-1, inmode=0, fpool=None, alg_kc=None):
        """Wm: (Cout, kh*kw*Cin) in (ky,kx,cin) order.  fpool = (ph, pw, kind) fuses a non-overlapping
        pool over ph*pw in {2,4} conv outputs into the epilogue; the OUT buffer then holds the pooled map."""
        h, w, cin = shape_in
        cout, K = Wm.shape
        assert K == kh * kw * cin
        fk = alg_kc if alg_kc is not None else K * cout      # algorithmic MACs per output pixel (channel padding excluded)
        kpad = -(-K // N.K_ALIGN) * N.K_ALIGN
        Wp = np.zeros((cout, kpad), np.float32)
        Wp[:, :K] = Wm
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_CONV
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = src, dst, res
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, cin
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = ho, wo, cout
        r[N.C_KH], r[N.C_KW], r[N.C_SH], r[N.C_SW], r[N.C_PT], r[N.C_PL] = kh, kw, sh, sw, pt, pl
        r[N.C_ACT] = act
        r[N.C_WOFF] = self.add_blob(Wp)
        r[N.C_BOFF] = self.add_blob(bias) if bias is not None else -1
        r[N.C_PSOFF] = self.add_blob(ps) if ps is not None else -1
        r[N.C_PTOFF] = self.add_blob(pt_) if pt_ is not None else -1
        r[N.C_INMODE] = inmode
        if fpool is not None:
            ph, pw, kind = fpool
            assert ph * pw in (2, 4) and res < 0 and ho // ph >= 1 and wo // pw >= 1
            r[N.C_FPOOLH], r[N.C_FPOOLW], r[N.C_POOLKIND] = ph, pw, kind
            self.flops += 2 * fk * (ho // ph * ph) * (wo // pw * pw)     # only the pooled region is computed
            ho, wo = ho // ph, wo // pw
        else:
            self.flops += 2 * fk * ho * wo
        self.rows.append(r)
        self.use_buf(dst, ho * wo * cout)
        return (ho, wo, cout)

    def dual(self, Wm_proj, b_proj, Wm_exp, b_exp):
        """Mark the last two rows -- a linear 1x1 projection and the in-place 1x1 expansion whose residual it is -- as fusable
        into ONE two-source GEMM (include/iss.h ISS_C_DUALW / ISS_C_DUALB): appends the concatenated matrix [W_exp | W_proj] and
        the summed bias to the blob.  The two rows stay as they are (the library falls back to them whenever it has to)."""
        rp, re = self.rows[-2], self.rows[-1]
        assert rp[N.C_OP] == re[N.C_OP] == N.OP_CONV and re[N.C_RES] == rp[N.C_OUT] == re[N.C_OUT] and rp[N.C_ACT] == 0
        assert Wm_exp.shape == (re[N.C_COUT], re[N.C_CIN]) and Wm_proj.shape == (rp[N.C_COUT], rp[N.C_CIN]) and rp[N.C_COUT] == re[N.C_COUT]
        if re[N.C_CIN] % N.K_ALIGN or rp[N.C_CIN] % N.K_ALIGN:
            return False
        re[N.C_DUALW] = 1 + self.add_blob(np.concatenate([Wm_exp, Wm_proj], axis=1))
        re[N.C_DUALB] = 1 + self.add_blob(np.asarray(b_exp, np.float32) + np.asarray(b_proj, np.float32))
        return True

    def pool(self, src, dst, shape_in, kh, kw, sh, sw, pt, pl, ho, wo, kind):
        h, w, c = shape_in
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_POOL
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = src, dst, -1
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, c
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = ho, wo, c
        r[N.C_KH], r[N.C_KW], r[N.C_SH], r[N.C_SW], r[N.C_PT], r[N.C_PL] = kh, kw, sh, sw, pt, pl
        r[N.C_POOLKIND] = kind
        for col in (N.C_WOFF, N.C_BOFF, N.C_PSOFF, N.C_PTOFF):
            r[col] = -1
        self.rows.append(r)
        self.use_buf(dst, ho * wo * c)
        return (ho, wo, c)

    def act(self, buf, shape_in, code, alpha):
        """Elementwise activation in place on `buf` (ISS_OP_ACT: elu / leaky relu / selu / softplus)."""
        h, w, c = shape_in
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_ACT
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = buf, buf, -1
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, c
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = h, w, c
        r[N.C_ACT] = code
        prm = tuple(alpha) if isinstance(alpha, tuple) else (alpha, 0.0, 0.0)      # code 9: (negative_slope, max_value, threshold)
        bits = np.array(prm, np.float32).view(np.int32)
        r[N.C_ACTPARAM], r[N.C_ACTPARAM2], r[N.C_ACTPARAM3] = int(bits[0]), int(bits[1]), int(bits[2])
        for col in (N.C_WOFF, N.C_BOFF, N.C_PSOFF, N.C_PTOFF):
            r[col] = -1
        self.rows.append(r)
        return shape_in

    def elt(self, kind, src, dst, shape_in, shape_out, res=-1, nch=0, soff=0, doff=0, perm=(0, 0, 0), relu=False):
        """Merge / data-movement row (ISS_OP_ELT, include/iss.h): binary kinds on (src, res) -> dst; COPY / ZERO of `nch` channels
        from channel soff of src to channel doff of dst; PERMUTE of the (H, W, C) axes."""
        h, w, c = shape_in
        ho, wo, co = shape_out
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_ELT
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = src, dst, res
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, c
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = ho, wo, co
        r[N.C_ACT] = kind
        r[N.C_ORDER] = 1 if relu else 0
        if kind in (N.ELT_COPY, N.ELT_ZERO):
            r[N.C_KH], r[N.C_PT], r[N.C_PL] = nch, soff, doff
        elif kind == N.ELT_PERMUTE:
            r[N.C_KH], r[N.C_KW], r[N.C_SH] = perm
        for col in (N.C_WOFF, N.C_BOFF, N.C_PSOFF, N.C_PTOFF):
            r[col] = -1
        self.rows.append(r)
        self.use_buf(dst, ho * wo * co)
        return shape_out

    def softmax(self, src, dst, shape_in):
        h, w, c = shape_in
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_SOFTMAX
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = src, dst, -1
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, c
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = h, w, c
        for col in (N.C_WOFF, N.C_BOFF, N.C_PSOFF, N.C_PTOFF):
            r[col] = -1
        self.rows.append(r)
        self.use_buf(dst, h * w * c)
        return shape_in

    def statpool(self, src, dst, shape_in):
        h, w, c = shape_in
        r = [0] * N.PROG_COLS
        r[N.C_OP] = N.OP_STATPOOL
        r[N.C_IN], r[N.C_OUT], r[N.C_RES] = src, dst, -1
        r[N.C_H], r[N.C_W], r[N.C_CIN] = h, w, c
        r[N.C_HO], r[N.C_WO], r[N.C_COUT] = 1, 1, 2 * c * h
        for col in (N.C_WOFF, N.C_BOFF, N.C_PSOFF, N.C_PTOFF):
            r[col] = -1
        self.rows.append(r)
        self.use_buf(dst, 2 * c * h)
        return (1, 1, 2 * c * h)

    def finish(self, in_shape, out_dim, patch_input):
        nbuf = max(self.buf_elems) + 1
        be = np.array([self.buf_elems.get(i, 1) for i in range(nbuf)], dtype=np.int64)
        blob = np.concatenate(self.blob) if self.blob else np.zeros(4, np.float32)
        return CompiledNet(np.array(self.rows, dtype=np.int32).reshape(-1, N.PROG_COLS), blob, be,
                           tuple(int(v) for v in in_shape), int(out_dim), float(self.flops), patch_input)


def _bn_affine(L):
    sc = (L['gamma'].astype(np.float64) / np.sqrt(L['var'].astype(np.float64) + L['eps']))
    sh = L['beta'].astype(np.float64) - L['mean'].astype(np.float64) * sc
    return sc, sh


CH_ALIGN = 32      # channel counts of intermediate activations are padded to this (the k-tile of the MFMA kernels)


def _can_fold_forward(layers, j, sc):
    """A BatchNorm BEHIND an activation (y = sc * r + sft per channel) can be moved into the NEXT linear layer -- W' = W diag(sc),
    b' = b + W sft -- when everything in between commutes with a per-channel affine map: dropout, flatten, average pools, max pools
    if every sc > 0 (max(sc r + sft) = sc max(r) + sft), and the consumer is a Dense layer or a Conv2D that does not zero-pad (the
    padding zeros of a 'same' convolution are zeros of y, not of r)."""
    for L in layers[j:]:
        ty = L['type']
        if ty in ('dropout', 'flatten', 'avgpool', 'globalavgpool'):
            if ty == 'avgpool' and L.get('padding', 'valid') == 'same':
                return False
            continue
        if ty in ('maxpool', 'globalmaxpool'):
            if not np.all(sc > 0):
                return False
            continue
        if ty == 'dense':
            return True
        if ty == 'conv2d':
            return L.get('padding', 'valid') == 'valid' and not L.get('pad')
        return False
    return False


def expand_generic_layers(layers):
    """Layers the op program has no kernel for, re-expressed through the ones it has -- slower than a dedicated kernel would be,
    but the same function (the reference's `keras.models.load_model`, segmenter.py:129-131, takes any model_config):
      * dilated Conv2D: the kernel with dilation - 1 zeros between its taps, i.e. an ordinary ((kh-1) dy + 1) x ((kw-1) dx + 1) filter
        (Keras pads 'same' for exactly that effective size);
      * DepthwiseConv2D (and the depthwise half of SeparableConv2D): an ordinary convolution whose (cin, cin * multiplier) tap
        matrices are zero except where output channel c * multiplier + m meets input channel c."""
    out = []
    for L in layers:
        L = dict(L)
        if L['type'] == 'depthwise':
            kh, kw, cin, mult = L['W'].shape
            W = np.zeros((kh, kw, cin, cin * mult), np.float32)
            for c in range(cin):
                W[:, :, c, c * mult:(c + 1) * mult] = L['W'][:, :, c, :]
            L['type'], L['W'], L['alg_macs'] = 'conv2d', W, kh * kw * cin * mult
        if L['type'] == 'conv2d' and tuple(L.get('dilation', (1, 1))) != (1, 1):
            L.setdefault('alg_macs', int(np.prod(L['W'].shape)))
            dy, dx = L['dilation']
            kh, kw, cin, cout = L['W'].shape
            W = np.zeros(((kh - 1) * dy + 1, (kw - 1) * dx + 1, cin, cout), np.float32)
            W[::dy, ::dx] = L['W']
            L['W'] = W
        L.pop('dilation', None)
        out.append(L)
    return out


class _Bufs:
    """Activation-buffer ids of a program under construction: the lowest id that holds no live tensor and is not the one being
    read.  A chain model ping-pongs between 0 and 1; a graph pins the tensors that still have readers."""

    def __init__(self):
        self.pinned = {}                            # buffer id -> readers left

    def alloc(self, cur):
        b = 0
        while b == cur or b in self.pinned:
            b += 1
        return b

    def pin(self, b, readers):
        if b >= 0 and readers > 0:
            self.pinned[b] = self.pinned.get(b, 0) + readers

    def release(self, b):
        if b in self.pinned:
            self.pinned[b] -= 1
            if self.pinned[b] <= 0:
                del self.pinned[b]


def _is_graph(layers):
    return any('inputs' in L for L in layers)


def compile_layers(layers, in_shape, patch_input=True, fuse_pool=True, pad_channels=True, fold_post_bn=True):
    """Lower a sequential layer list onto the op program.  Fusions: conv/dense + bias,
    + BatchNorm directly after (folded into W, b), + relu/sigmoid/tanh, + BatchNorm after the
    activation (epilogue scale/shift), + a non-overlapping 'valid' max/avg pool over 2 or 4 outputs.
    Anything left over becomes an identity 1x1 conv.

    fold_post_bn: `Conv2D(activation='relu')` + `BatchNormalization()` -- the BatchNorm BEHIND the activation -- is folded into the
    next Dense / unpadded Conv2D where that is exact (`_can_fold_forward`) instead of becoming an epilogue affine of its producer: the
    program then has the same shape as a conv - BN - relu net and takes the same kernels (a post-activation affine keeps a layer off
    the shared-first-layer / one-wave-per-SIMD forms).  Float64 on the host; results move by float32 rounding of a re-associated sum.

    pad_channels: an intermediate activation whose channel count is not a multiple of 32 (48, 96, 20 ...) is stored with
    zero channels appended (zero weight rows / bias / scale in the producer, zero weight columns in every consumer), so
    that every conv / dense behind the first layer runs on the vectorised MFMA kernels (Cin % 32 == 0) instead of the
    scalar-gather path (only from 16 channels up: below that the padding would more than double the work); results are unchanged (the extra products are exact zeros) and `flops_per_sample` keeps counting
    the model's own MACs."""

    layers = expand_generic_layers(layers)          # dilated / depthwise convolutions as ordinary ones on zero-filled kernels
    B = _Builder()
    opts = dict(patch_input=patch_input, fuse_pool=fuse_pool, pad_channels=pad_channels, fold_post_bn=fold_post_bn)
    shape = tuple(int(v) for v in in_shape)
    if _is_graph(layers):
        cur, shape, pmap = _compile_graph(B, layers, shape, opts)
    else:
        cur, shape, pmap = _compile_chain(B, _Bufs(), layers, N.BUF_INPUT, shape, np.arange(shape[2]), True, False, opts)
    if cur == N.BUF_INPUT:
        raise ValueError("empty network")
    if len(pmap) != shape[2]:
        raise NotImplementedError("network output is channel-padded")
    out_dim = shape[0] * shape[1] * shape[2]
    return B.finish(in_shape, out_dim, patch_input)


def _absorbs_padding(layers, j, at_end):
    """Whether the tensor produced in front of layers[j] may carry zero padding channels: the next layer that is not transparent to
    them (pools, flatten, dropout are) rebuilds the channel axis from its weights (conv / dense, or the identity carrier of a
    stand-alone BatchNorm / activation).  Softmax, Reshape, Permute and the merge layers need the model's own channels."""
    for L in layers[j:]:
        ty = L['type']
        if ty in ('dropout', 'flatten', 'maxpool', 'avgpool', 'globalavgpool', 'globalmaxpool'):
            continue
        return ty in ('conv2d', 'dense', 'batchnorm') or (ty == 'activation' and L['fn'] != 'softmax')
    return at_end


def _compile_chain(B, bufs, layers, cur, shape, pmap, first, padded_end, opts):
    """Lower a single-input single-output run of layers reading buffer `cur` (logical `shape`, physical channel map `pmap`); returns
    (buffer, shape, pmap) of its result.  `first`: the run starts at the network input.  `padded_end`: whoever reads the result
    accepts padding channels (_absorbs_padding)."""
    patch_input, fuse_pool, pad_channels, fold_post_bn = (opts[k] for k in ('patch_input', 'fuse_pool', 'pad_channels', 'fold_post_bn'))
    # pmap: physical channel -> logical channel of the current activation (-1 = padding)
    i, n = 0, len(layers)
    carry = None                                    # (sc, sft) float64 per LOGICAL input feature of the next linear layer (fold_post_bn)
    src0 = cur                                      # the run's own input: never overwritten, released by the caller

    def nxt_buf():
        return bufs.alloc(cur)

    def peek(j):
        while j < n and layers[j]['type'] == 'dropout':
            j += 1
        return j

    while i < n:
        L = layers[i]
        ty = L['type']
        if ty == 'dropout':
            i += 1
            continue
        if ty == 'flatten':
            hw, c = shape[0] * shape[1], shape[2]
            pmap = (np.where(pmap >= 0, pmap, -(1 << 40))[None, :] + (np.arange(hw) * c)[:, None]).ravel()
            pmap = np.where(pmap >= 0, pmap, -1)
            shape = (1, 1, hw * c)
            if carry is not None:                   # feature index = position * channels + channel
                carry = (np.tile(carry[0], hw), np.tile(carry[1], hw))
            i += 1
            continue
        if ty in ('conv2d', 'dense', 'batchnorm', 'activation') and not (ty == 'activation' and L['fn'] == 'softmax'):
            h, w, cin = shape
            if ty == 'conv2d':
                W = L['W']
                kh, kw, wc, cout = W.shape
                assert wc == cin, (L.get('name'), wc, cin)
                sh, sw = L.get('strides', (1, 1))
                if L.get('padding', 'valid') == 'same':
                    assert not L.get('pad'), "explicit padding goes with padding='valid'"
                    ho, pt = _same_pads(h, kh, sh)
                    wo, pl = _same_pads(w, kw, sw)
                elif L.get('pad'):                  # ZeroPadding2D merged into this convolution: (top, bottom, left, right)
                    zt, zb, zl, zr = L['pad']
                    ho, wo, pt, pl = (h + zt + zb - kh) // sh + 1, (w + zl + zr - kw) // sw + 1, zt, zl
                else:
                    ho, wo, pt, pl = (h - kh) // sh + 1, (w - kw) // sw + 1, 0, 0
                Wm = W.transpose(3, 0, 1, 2).reshape(cout, -1).astype(np.float64)
                bias = None if L.get('b') is None else L['b'].astype(np.float64)
                act_name = L.get('activation', 'linear')
                j = i + 1
            elif ty == 'dense':
                assert h == 1 and w == 1, "Dense on un-flattened input is not supported"
                W = L['W']
                assert W.shape[0] == cin, (L.get('name'), W.shape, cin)
                cout = W.shape[1]
                kh = kw = sh = sw = 1
                ho = wo = 1
                pt = pl = 0
                Wm = W.T.astype(np.float64)
                bias = None if L.get('b') is None else L['b'].astype(np.float64)
                act_name = L.get('activation', 'linear')
                j = i + 1
            else:                                   # stand-alone BN / activation: identity 1x1 conv carrier
                cout = cin
                kh = kw = sh = sw = 1
                ho, wo, pt, pl = h, w, 0, 0
                Wm = np.eye(cin, dtype=np.float64)
                bias = None
                act_name = 'linear'
                j = i
            if carry is not None:                   # the producer's post-activation BatchNorm, moved here (see _can_fold_forward)
                assert ty in ('conv2d', 'dense') and pt == 0 and pl == 0 and len(carry[0]) == cin, (ty, pt, pl, len(carry[0]), cin)
                W3 = Wm.reshape(cout, kh * kw, cin)
                bias = (bias if bias is not None else np.zeros(cout)) + (W3 * carry[1][None, None, :]).sum(axis=(1, 2))
                Wm = (W3 * carry[0][None, None, :]).reshape(cout, kh * kw * cin)
                carry = None
            softmax_after = False
            if act_name == 'softmax':
                act_name, softmax_after = 'linear', True
            # BN straight after the linear part -> fold
            j = peek(j)
            if act_name in (None, 'linear') and not softmax_after and j < n and layers[j]['type'] == 'batchnorm':
                sc, sft = _bn_affine(layers[j])
                Wm = Wm * sc[:, None]
                bias = (bias if bias is not None else 0.0) * sc + sft
                j = peek(j + 1)
            # activation
            act_alpha = L.get('alpha') if ty in ('conv2d', 'dense') else None
            if act_name in (None, 'linear') and not softmax_after and j < n and layers[j]['type'] == 'activation' \
                    and layers[j]['fn'] in ('relu', 'sigmoid', 'tanh', 'elu', 'leaky_relu', 'selu', 'softplus', 'relu_max', 'relu_general'):
                act_name = layers[j]['fn']
                act_alpha = layers[j].get('alpha')
                j = peek(j + 1)
            # elu / leaky relu / selu / softplus are not fused into the producer's epilogue: the conv runs linear (with a fused MAX pool
            # in front of the activation when there is one: these functions are non-decreasing for alpha >= 0, so they commute with it)
            # and an elementwise ISS_OP_ACT row follows; a BatchNorm behind such an activation is lowered on its own afterwards
            act_op = None
            if act_name in _ACT_OP_CODE:
                if act_name == 'relu_general':           # (negative_slope, max_value, threshold)
                    alpha = tuple(float(v) for v in act_alpha)
                else:
                    alpha = float(act_alpha if act_alpha is not None else _ACT_DEFAULT_ALPHA.get(act_name, 0.0))
                act_op = (_ACT_OP_CODE[act_name], alpha)
                act_name = 'linear'
            if act_name not in _ACT_CODE:
                raise NotImplementedError(f"activation {act_name!r}")
            # BN after the activation -> epilogue affine
            ps = pt_ = None
            if act_op is None and not softmax_after and j < n and layers[j]['type'] == 'batchnorm':
                sc, sft = _bn_affine(layers[j])
                if fold_post_bn and _can_fold_forward(layers, j + 1, sc):
                    carry = (sc, sft)               # into the next linear layer's weights and bias
                else:
                    ps, pt_ = sc.astype(np.float32), sft.astype(np.float32)
                j = peek(j + 1)
            # non-overlapping 'valid' pool of 2 or 4 outputs right after -> epilogue
            fpool = None
            if fuse_pool and not softmax_after and j < n and layers[j]['type'] in ('maxpool', 'avgpool') and \
                    (act_op is None or (layers[j]['type'] == 'maxpool' and (act_op[1][0] if isinstance(act_op[1], tuple) else act_op[1]) >= 0.0)):
                PL = layers[j]
                pph, ppw = PL['pool']
                pst = tuple(PL.get('strides') or PL['pool'])
                if PL.get('padding', 'valid') == 'valid' and pst == (pph, ppw) and pph * ppw in (2, 4) \
                        and ho // pph >= 1 and wo // ppw >= 1:
                    fpool = (pph, ppw, 0 if PL['type'] == 'maxpool' else 1)
                    j = peek(j + 1)
            if j == i:                               # nothing consumed (cannot happen) -> avoid a loop
                raise RuntimeError("lowering made no progress")
            dst = nxt_buf()
            inmode = 1 if (first and patch_input) else 0
            if inmode == 1 and not (cin == 1 and h == 68 and w <= 24):
                raise ValueError(f"patch input must be (68, <=24, 1); got {shape}")
            # physical operand: input channels follow `pmap` (zero columns for padding channels); output channels are
            # padded when another conv / dense consumes them
            alg_kc = L.get('alg_macs') or Wm.shape[0] * Wm.shape[1]      # (zero-filled generic kernels: the model's own MACs)
            cin_p = len(pmap)
            if cin_p != cin or np.any(pmap != np.arange(cin)):
                W3 = Wm.reshape(cout, kh * kw, cin)
                Wp3 = np.zeros((cout, kh * kw, cin_p))
                Wp3[:, :, pmap >= 0] = W3[:, :, pmap[pmap >= 0]]
                Wm = Wp3.reshape(cout, kh * kw * cin_p)
            later = _absorbs_padding(layers, j, padded_end)
            cout_p = cout
            if pad_channels and later and not softmax_after and cout % CH_ALIGN and cout >= CH_ALIGN // 2:   # at most 2x the work
                cout_p = -(-cout // CH_ALIGN) * CH_ALIGN
                Wm = np.concatenate((Wm, np.zeros((cout_p - cout, Wm.shape[1]))), axis=0)
                zpad = np.zeros(cout_p - cout)
                bias = None if bias is None else np.concatenate((np.asarray(bias, np.float64), zpad))
                if ps is not None:
                    ps, pt_ = np.concatenate((ps, zpad)).astype(np.float32), np.concatenate((pt_, zpad)).astype(np.float32)
                elif _ACT_CODE[act_name] >= 2:               # sigmoid(0) / tanh(0): keep padding channels at exactly 0
                    ps = np.concatenate((np.ones(cout), zpad)).astype(np.float32)
                    pt_ = np.zeros(cout_p, np.float32)
            pshape = B.conv(cur, dst, (h, w, cin_p), Wm.astype(np.float32), kh, kw, sh, sw, pt, pl, ho, wo,
                            bias=None if bias is None else np.asarray(bias, np.float32),
                            act=_ACT_CODE[act_name], ps=ps, pt_=pt_, inmode=inmode, fpool=fpool, alg_kc=alg_kc)
            if act_op is not None:
                B.act(dst, (pshape[0], pshape[1], cout_p), act_op[0], act_op[1])
            shape = (pshape[0], pshape[1], cout)
            pmap = np.concatenate((np.arange(cout), np.full(cout_p - cout, -1)))
            cur, first = dst, False
            if softmax_after:
                dst = nxt_buf()
                shape = B.softmax(cur, dst, shape)
                cur = dst
            i = j
            continue
        if first and patch_input:                   # network does not start with a conv: identity carrier
            dst = nxt_buf()
            shape = B.conv(cur, dst, shape, np.eye(1, dtype=np.float32), 1, 1, 1, 1, 0, 0, shape[0], shape[1], inmode=1)
            cur, first = dst, False
        pshape = (shape[0], shape[1], len(pmap))    # pools / softmax work on the stored (padded) channel axis
        if ty == 'activation' and L['fn'] == 'softmax':
            if len(pmap) != shape[2]:
                raise NotImplementedError("softmax over a channel-padded activation")
            dst = nxt_buf()
            shape = B.softmax(cur, dst, shape)
            cur = dst
        elif ty in ('maxpool', 'avgpool'):
            h, w, c = shape
            ph, pw = L['pool']
            sh, sw = L.get('strides') or L['pool']
            if L.get('padding', 'valid') == 'same':     # (average: the mean leaves the padding out, as Keras / TF do)
                ho, pt = _same_pads(h, ph, sh)
                wo, pl = _same_pads(w, pw, sw)
            else:
                ho, wo, pt, pl = (h - ph) // sh + 1, (w - pw) // sw + 1, 0, 0
            dst = nxt_buf()
            B.pool(cur, dst, pshape, ph, pw, sh, sw, pt, pl, ho, wo, 0 if ty == 'maxpool' else 1)
            shape = (ho, wo, c)
            cur = dst
        elif ty in ('globalavgpool', 'globalmaxpool'):
            h, w, c = shape
            dst = nxt_buf()
            B.pool(cur, dst, pshape, h, w, h, w, 0, 0, 1, 1, 1 if ty == 'globalavgpool' else 0)
            shape = (1, 1, c)
            cur = dst
        elif ty == 'reshape':                       # row-major on (H, W, C) storage: a new shape for the same floats, no row
            if len(pmap) != shape[2]:
                raise NotImplementedError("internal: Reshape of a channel-padded activation")
            shape = _reshape_target(L['target'], shape)
            pmap = np.arange(shape[2])
        elif ty == 'permute':
            if len(pmap) != shape[2]:
                raise NotImplementedError("internal: Permute of a channel-padded activation")
            perm = tuple(int(v) for v in L['perm'])
            if perm != (0, 1, 2):
                oshape = tuple(shape[k] for k in perm)
                dst = nxt_buf()
                B.elt(N.ELT_PERMUTE, cur, dst, shape, oshape, perm=perm)
                shape, cur = oshape, dst
                pmap = np.arange(shape[2])
        else:
            raise NotImplementedError(ty)
        first = False
        i += 1
    assert carry is None, "a deferred BatchNorm was never consumed"
    return cur, shape, pmap


def _reshape_target(target, shape):
    """keras.layers.Reshape(target_shape) on a stored (H, W, C) activation: (n,) -> (1, 1, n); (a, b, c) as it is; one -1 is inferred."""
    total = int(shape[0]) * int(shape[1]) * int(shape[2])
    t = [int(v) for v in target]
    if t.count(-1) > 1 or len(t) not in (1, 3):
        raise NotImplementedError(f"Reshape to {tuple(target)}: the op program stores (H, W, C) or flat activations")
    if -1 in t:
        known = int(np.prod([v for v in t if v != -1])) if len(t) > 1 else 1
        t[t.index(-1)] = total // max(known, 1)
    if int(np.prod(t)) != total:
        raise ValueError(f"Reshape {tuple(shape)} -> {tuple(target)}: element counts differ")
    return (1, 1, t[0]) if len(t) == 1 else tuple(t)


_MERGE_KIND = {'add': N.ELT_ADD, 'subtract': N.ELT_SUB, 'multiply': N.ELT_MUL, 'maximum': N.ELT_MAX, 'minimum': N.ELT_MIN,
               'average': N.ELT_AVG}
_MERGES = tuple(_MERGE_KIND) + ('concatenate',)
GRAPH_INPUT = '__input__'


def _compile_graph(B, layers, in_shape, opts):
    """Lower a layer list whose entries name their producers ('inputs': [layer names], GRAPH_INPUT = the network input) -- what a
    functional Keras model that is not a chain parses to.  Runs of single-input layers between branch and merge points go through
    `_compile_chain` (with all its fusions); Add / Subtract / Multiply / Average / Maximum / Minimum / Concatenate become ISS_OP_ELT
    rows (one-thread-per-element kernels: a graph costs speed, not the run).  Buffers are handed out by liveness (`_Bufs`)."""
    names = [L.get('name') for L in layers]
    if None in names or len(set(names)) != len(names) or GRAPH_INPUT in names:
        raise ValueError("a graph-shaped layer list needs unique layer names")
    idx_of = {nm: k for k, nm in enumerate(names)}
    ins = []
    for k, L in enumerate(layers):
        src = L.get('inputs')
        if src is None:
            src = [names[k - 1] if k else GRAPH_INPUT]
        for nm in src:
            if nm != GRAPH_INPUT and nm not in idx_of:
                raise ValueError(f"layer {L['name']!r} reads {nm!r}, which is not a layer of the model")
        if L['type'] not in _MERGES and len(src) != 1:
            raise ValueError(f"layer {L['name']!r} ({L['type']}) takes one input, got {len(src)}")
        ins.append(list(src))
    readers = {nm: [] for nm in names + [GRAPH_INPUT]}
    for k, src in enumerate(ins):
        for nm in src:
            readers[nm].append(k)
    sinks = [nm for nm in names if not readers[nm]]
    if len(sinks) != 1:
        raise NotImplementedError(f"a model with {len(sinks)} outputs ({sinks}): the segmenter networks have one")
    # topological order that keeps the list's own order where it can
    order, placed, left = [], {GRAPH_INPUT}, list(range(len(layers)))
    while left:
        ready = [k for k in left if all(nm in placed for nm in ins[k])]
        if not ready:
            raise ValueError("the layer graph has a cycle")
        order.append(ready[0]); placed.add(names[ready[0]]); left.remove(ready[0])

    def run_from(k):
        """The run of single-input layers that starts at layer k and can be lowered as one chain."""
        run = [k]
        while True:
            rd = readers[names[run[-1]]]
            if len(rd) != 1 or layers[rd[0]]['type'] in _MERGES or len(ins[rd[0]]) != 1:
                return run
            run.append(rd[0])

    def accepts_padding(nm):
        rd = readers[nm]
        return bool(rd) and all(layers[k]['type'] not in _MERGES and _absorbs_padding([layers[q] for q in run_from(k)], 0, False) for k in rd)

    bufs = _Bufs()
    tensors = {GRAPH_INPUT: (N.BUF_INPUT, tuple(in_shape), np.arange(in_shape[2]))}
    at_input = True                                  # the network input has not been copied into an activation buffer
    rd0 = readers[GRAPH_INPUT]
    if not rd0:
        raise ValueError("no layer reads the network input")
    if len(rd0) > 1 or layers[rd0[0]]['type'] in _MERGES or not opts['patch_input']:
        if opts['patch_input']:                      # several readers (or a merge): one identity carrier materialises the patch
            b = bufs.alloc(N.BUF_INPUT)
            shp = B.conv(N.BUF_INPUT, b, tuple(in_shape), np.eye(1, dtype=np.float32), 1, 1, 1, 1, 0, 0, in_shape[0], in_shape[1], inmode=1)
            tensors[GRAPH_INPUT] = (b, shp, np.arange(shp[2]))
            bufs.pin(b, len(rd0))
            at_input = False
        elif any(layers[k]['type'] in _MERGES for k in rd0):
            raise NotImplementedError("a merge layer directly on the network input of a feature-vector model")
    done = set()
    for k in order:
        if k in done:
            continue
        L = layers[k]
        if L['type'] in _MERGES:
            ts = [tensors[nm] for nm in ins[k]]
            for nm, (b, shp, pm) in zip(ins[k], ts):
                if len(pm) != shp[2]:
                    raise NotImplementedError(f"internal: merge input {nm!r} is channel-padded")
            out_nm, nrd = names[k], len(readers[names[k]])
            if L['type'] == 'concatenate':
                nd = 3 if any(t[1][0] * t[1][1] > 1 for t in ts) else 1          # stored rank: (H, W, C) or flat
                ax = int(L.get('axis', -1))
                ax = ax + nd + 1 if ax < 0 else ax                                # Keras counts the batch axis
                if not 1 <= ax <= nd:
                    raise ValueError(f"Concatenate axis {L.get('axis')} on rank-{nd} activations")
                ax = ax - 1 + (3 - nd)                                            # 0 = H, 1 = W, 2 = C of the stored shape
                base = ts[0][1]
                for t in ts:
                    if any(t[1][d] != base[d] for d in range(3) if d != ax):
                        raise ValueError(f"Concatenate ({out_nm}): input shapes {[t_[1] for t_ in ts]} differ off axis {ax}")
                tot = sum(t[1][ax] for t in ts)
                oshape = tuple(tot if d == ax else base[d] for d in range(3))
                # COPY rows on a re-read grid: the axis in front of `ax` are pixels, everything from `ax` on is one "channel" run
                grid = int(np.prod(base[:ax])) if ax else 1
                inner = int(np.prod(base[ax + 1:])) if ax < 2 else 1
                ctot = tot * inner
                cp = ctot
                if ax == 2 and opts['pad_channels'] and accepts_padding(out_nm) and ctot % CH_ALIGN and ctot >= CH_ALIGN // 2:
                    cp = -(-ctot // CH_ALIGN) * CH_ALIGN
                dst = bufs.alloc(-1)
                off = 0
                for (b, shp, pm) in ts:
                    nch = shp[ax] * inner
                    B.elt(N.ELT_COPY, b, dst, (grid, 1, nch), (grid, 1, cp), nch=nch, soff=0, doff=off)
                    off += nch
                if cp > ctot:
                    B.elt(N.ELT_ZERO, dst, dst, (grid, 1, cp), (grid, 1, cp), nch=cp - ctot, doff=ctot)
                res = (dst, oshape, np.concatenate((np.arange(ctot), np.full(cp - ctot, -1))) if ax == 2 else np.arange(oshape[2]))
            else:
                base = ts[0][1]
                if any(t[1] != base for t in ts) or len(ts) < 2:
                    raise ValueError(f"{L['type']} ({out_nm}): needs two or more inputs of one shape, got {[t_[1] for t_ in ts]}")
                if L['type'] == 'subtract' and len(ts) != 2:
                    raise ValueError("Subtract takes exactly two inputs")
                kind = _MERGE_KIND[L['type']]
                many_avg = L['type'] == 'average' and len(ts) > 2
                # a ReLU that is the merge's only reader (the Add + ReLU of a residual block) rides in the last merge row
                rd = readers[out_nm]
                fuse = (not many_avg and len(rd) == 1 and layers[rd[0]]['type'] == 'activation' and layers[rd[0]]['fn'] == 'relu'
                        and ins[rd[0]] == [out_nm])
                dst = bufs.alloc(-1)
                B.elt(N.ELT_ADD if many_avg else kind, ts[0][0], dst, base, base, res=ts[1][0], relu=fuse and len(ts) == 2)
                for q, t in enumerate(ts[2:]):
                    B.elt(N.ELT_ADD if many_avg else kind, dst, dst, base, base, res=t[0], relu=fuse and q == len(ts) - 3)
                if fuse:                              # the activation layer is done: its readers read the merge's buffer
                    done.add(rd[0])
                    out_nm = names[rd[0]]
                    nrd = len(readers[out_nm])
                if many_avg:                          # mean of n > 2 tensors: the sum through an identity 1x1 conv scaled by 1 / n
                    bufs.pin(dst, 1)
                    d2 = bufs.alloc(dst)
                    B.conv(dst, d2, base, np.eye(base[2], dtype=np.float64) / len(ts), 1, 1, 1, 1, 0, 0, base[0], base[1], alg_kc=base[2])
                    bufs.release(dst)
                    dst = d2
                res = (dst, base, np.arange(base[2]))
            for (b, shp, pm) in ts:
                bufs.release(b)
            tensors[out_nm] = res
            bufs.pin(res[0], nrd)
            continue
        run = run_from(k)
        done.update(run)
        src_nm = ins[k][0]
        b, shp, pm = tensors[src_nm]
        bufs.release(b)
        first = at_input and src_nm == GRAPH_INPUT
        last_nm = names[run[-1]]
        res = _compile_chain(B, bufs, [layers[q] for q in run], b, shp, pm, first, accepts_padding(last_nm), opts)
        tensors[last_nm] = res
        bufs.pin(res[0], len(readers[last_nm]))
    return tensors[sinks[0]]


# ------------------------------------------------------------------------------ ResNet-101
def compile_resnet101(params, feat_dim=64, frames=144, eps=1e-5, window_input=False):
    """params: dict keyed like resnet.py's state_dict (conv OIHW, BN weight/bias/running_*).
    Lowers resnet.py:78-130 (Bottleneck [3,4,23,3], m_channels 32) for a fixed number of
    frames; BatchNorm (eval mode) is folded into the preceding conv.  window_input=True makes the first conv read
    its (feat_dim, frames) image straight from the resident (T, feat_dim) vbx features (iss_vbx_embed)."""
    B = _Builder()

    def fold(conv, bn):
        W = np.asarray(params[conv + '.weight'], np.float64)                      # OIHW
        if (bn + '.weight') not in params:            # an export with BatchNorm already folded into the conv (onnx_reader.py)
            Wm = W.transpose(0, 2, 3, 1).reshape(W.shape[0], -1)
            return Wm.astype(np.float32), np.asarray(params[conv + '.bias'], np.float32), W.shape[2], W.shape[3]
        sc = np.asarray(params[bn + '.weight'], np.float64) / np.sqrt(np.asarray(params[bn + '.running_var'], np.float64) + eps)
        sft = np.asarray(params[bn + '.bias'], np.float64) - np.asarray(params[bn + '.running_mean'], np.float64) * sc
        Wm = (W * sc[:, None, None, None]).transpose(0, 2, 3, 1).reshape(W.shape[0], -1)
        return Wm.astype(np.float32), sft.astype(np.float32), W.shape[2], W.shape[3]

    def conv(src, dst, shape, cname, bname, stride, pad, act, res=-1, inmode=0):
        Wm, b, kh, kw = fold(cname, bname)
        h, w, _ = shape
        ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        return B.conv(src, dst, shape, Wm, kh, kw, stride, stride, pad, pad, ho, wo, bias=b, act=act, res=res, inmode=inmode)

    shape = (feat_dim, frames, 1)                     # NHWC view of torch's (B,1,F,T)
    A, Bb, Cc, D = 0, 1, 2, 3
    shape = conv(N.BUF_INPUT, A, shape, 'conv1', 'bn1', 1, 1, 1, inmode=2 if window_input else 0)
    cur = A
    spare = D
    for li, (planes, nblocks, stride) in enumerate(zip((32, 64, 128, 256), (3, 4, 23, 3), (1, 2, 2, 2)), 1):
        for bi in range(nblocks):
            s = stride if bi == 0 else 1
            p = f'layer{li}.{bi}'
            s1 = conv(cur, Bb, shape, p + '.conv1', p + '.bn1', 1, 0, 1)
            s2 = conv(Bb, Cc, s1, p + '.conv2', p + '.bn2', s, 1, 1)
            if (p + '.shortcut.0.weight') in params:
                conv(cur, spare, shape, p + '.shortcut.0', p + '.shortcut.1', s, 0, 0)
                shape = conv(Cc, spare, s2, p + '.conv3', p + '.bn3', 1, 0, 1, res=spare)   # in-place add + relu
                Wp, bp, _, _ = fold(p + '.shortcut.0', p + '.shortcut.1')                  # both as ONE GEMM over [conv2 out | block input]
                We, be, _, _ = fold(p + '.conv3', p + '.bn3')
                B.dual(Wp, bp, We, be)
                cur, spare = spare, cur
            else:
                shape = conv(Cc, cur, s2, p + '.conv3', p + '.bn3', 1, 0, 1, res=cur)
    shape = B.statpool(cur, Bb, shape)
    Wm = np.asarray(params['embedding.weight'], np.float32)
    B.conv(Bb, Cc, shape, Wm, 1, 1, 1, 1, 0, 0, 1, 1, bias=np.asarray(params['embedding.bias'], np.float32))
    return B.finish((feat_dim, frames, 1), Wm.shape[0], False)


# ------------------------------------------------------------------------------ synthetic nets
_STANDIN_HEADS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'standin_heads.npz')
# (nmel, nclasses, seed) of the three networks Segmenter(models='synthetic') builds -> key in standin_heads.npz
_STANDIN_KEYS = {(21, 3, 1): 'smn', (21, 2, 1): 'sm', (24, 2, 2): 'gender'}


def standin_head(nmel, nclasses, seed):
    """(W (128, C), b (C,)) of the calibrated last layer for the stand-ins Segmenter(models='synthetic') uses, or None.
    Fitted once on the CPU oracle by tests/golden/make_standin_heads.py (ridge least squares on the bench generator's ground
    truth and on the labels of the reference's golden CSVs for media/musanmix.wav) so that the stand-ins DECIDE: a random
    head answers one class on > 99.8 % of all slots, which turns every label test into a test of the energy detector."""
    key = _STANDIN_KEYS.get((nmel, nclasses, seed))
    if key is None or not os.path.exists(_STANDIN_HEADS):
        return None
    z = np.load(_STANDIN_HEADS)
    return np.asarray(z[key + '_W'], np.float32), np.asarray(z[key + '_b'], np.float32)


def synthetic_ina_like(nmel, nclasses, seed=0, head='auto'):
    """A seeded stand-in for the un-vendored Keras CNNs: same I/O contract
    ((68,nmel,1) -> nclasses softmax, segmenter.py:146-163,184-204) and the ~1.25 M
    parameter budget the reference quotes (Dockerfile:18), topology chosen here:
    4 x [conv-BN-relu] with two max-pools, two dense layers, softmax.
    head: 'auto' -> the calibrated last layer of `standin_head` when one exists for (nmel, nclasses, seed), else the seeded
    random one; None -> always the random one; (W, b) -> that one."""
    rng = np.random.default_rng(seed)

    def conv(kh, kw, cin, cout):
        return dict(type='conv2d', W=(rng.normal(0, np.sqrt(2.0 / (kh * kw * cin)), (kh, kw, cin, cout))).astype(np.float32),
                    b=rng.normal(0, 0.05, cout).astype(np.float32), strides=(1, 1), padding='valid', activation='linear')

    def bn(c):
        return dict(type='batchnorm', gamma=rng.uniform(0.8, 1.2, c).astype(np.float32),
                    beta=rng.normal(0, 0.1, c).astype(np.float32), mean=rng.normal(0, 0.1, c).astype(np.float32),
                    var=rng.uniform(0.5, 1.5, c).astype(np.float32), eps=1e-3)

    def dense(i, o, act):
        return dict(type='dense', W=rng.normal(0, np.sqrt(2.0 / i), (i, o)).astype(np.float32),
                    b=rng.normal(0, 0.05, o).astype(np.float32), activation=act)

    relu = dict(type='activation', fn='relu')
    h, w = 68, nmel
    L = [conv(4, 5, 1, 64), bn(64), relu]; h, w = h - 3, w - 4
    L += [conv(5, 3, 64, 64), bn(64), relu]; h, w = h - 4, w - 2
    L += [dict(type='maxpool', pool=(2, 2), strides=(2, 2), padding='valid')]; h, w = h // 2, w // 2
    L += [conv(3, 3, 64, 128), bn(128), relu]; h, w = h - 2, w - 2
    L += [conv(3, 3, 128, 128), bn(128), relu]; h, w = h - 2, w - 2
    L += [dict(type='maxpool', pool=(2, 1), strides=(2, 1), padding='valid')]; h = h // 2
    L += [dict(type='flatten'), dense(h * w * 128, 192, 'linear'), bn(192), relu, dict(type='dropout'),
          dense(192, 128, 'relu'), dense(128, nclasses, 'softmax')]
    if isinstance(head, str) and head == 'auto':
        head = standin_head(nmel, nclasses, seed)
    if head is not None:
        W, b = head
        assert W.shape == (128, nclasses) and b.shape == (nclasses,)
        L[-1] = dict(type='dense', W=np.asarray(W, np.float32), b=np.asarray(b, np.float32), activation='softmax')
    return L, (68, nmel, 1)


def synthetic_resnet101(seed=0):
    """Seeded, numerically tame ResNet-101 parameters keyed like resnet.py's state_dict (test / bench stand-in)."""
    rng = np.random.default_rng(seed)
    params = {}

    def conv(name, cout, cin, k):
        params[name + '.weight'] = rng.normal(0, np.sqrt(1.0 / (cin * k * k)), (cout, cin, k, k)).astype(np.float32)

    def bn(name, c):
        params[name + '.weight'] = rng.uniform(0.8, 1.2, c).astype(np.float32)
        params[name + '.bias'] = rng.normal(0, 0.1, c).astype(np.float32)
        params[name + '.running_mean'] = rng.normal(0, 0.1, c).astype(np.float32)
        params[name + '.running_var'] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    m = 32
    conv('conv1', m, 1, 3); bn('bn1', m)
    inp = m
    for li, (planes, nblocks, stride) in enumerate(zip((m, 2 * m, 4 * m, 8 * m), (3, 4, 23, 3), (1, 2, 2, 2)), 1):
        for bi in range(nblocks):
            p = f'layer{li}.{bi}'
            conv(p + '.conv1', planes, inp, 1); bn(p + '.bn1', planes)
            conv(p + '.conv2', planes, planes, 3); bn(p + '.bn2', planes)
            conv(p + '.conv3', 4 * planes, planes, 1); bn(p + '.bn3', 4 * planes)
            if (stride if bi == 0 else 1) != 1 or inp != 4 * planes:
                conv(p + '.shortcut.0', 4 * planes, inp, 1); bn(p + '.shortcut.1', 4 * planes)
            inp = 4 * planes
    params['embedding.weight'] = rng.normal(0, np.sqrt(1.0 / 16384), (256, 16384)).astype(np.float32)
    params['embedding.bias'] = rng.normal(0, 0.05, 256).astype(np.float32)
    return params
