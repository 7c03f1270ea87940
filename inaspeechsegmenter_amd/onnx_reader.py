"""Minimal ONNX reader for the x-vector ResNet-101 (`final.onnx`, remote_utils.py:13; loaded by the reference with
onnxruntime, vbx_segmenter.py:249-266).  No `onnx` / `onnxruntime` / protobuf package is needed: the file is a protobuf
message and only a handful of fields matter here, so this module walks the wire format itself.

    ModelProto.graph (7) -> GraphProto.node (1), .initializer (5)
    NodeProto: input (1), output (2), name (3), op_type (4), attribute (5)
    AttributeProto: name (1), f (2), i (3), ints (8)
    TensorProto: dims (1), data_type (2), float_data (4), int64_data (7), name (8), raw_data (9), double_data (10)

`load_resnet101_params(path)` returns a dict keyed like resnet.py's state_dict ('conv1.weight', 'layer3.7.bn2.running_var',
'embedding.weight', ...) -- what `keras_model.compile_resnet101` lowers.  Initializer NAMES are not relied on (constant folding
renames them to `onnx::Conv_123`) and neither is the node ORDER (conv3 and the shortcut projection of a stage's first block
have the same weight shape): the Conv nodes are assigned to resnet.py:48-75,105-135's convolutions by walking the graph
(`_order_convs_by_connectivity`), then checked against the expected (out, in, kh, kw), strides, pads and group.  Two export styles are understood:
BatchNormalization nodes kept (their four tensors become bnX.weight / bias / running_mean / running_var) or folded into the
convolutions (Conv carries a bias: returned as '<conv>.bias' with no bn entries; compile_resnet101 then uses it as is)."""
import struct

import numpy as np


# ------------------------------------------------------------------------------ protobuf wire format
def _varint(buf, pos):
    r = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, pos
        shift += 7
        if shift > 70:
            raise ValueError('malformed varint')


def _fields(buf):
    """yield (field_number, wire_type, value) of one message; value: int (varint / fixed) or memoryview (length-delimited)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        if pos > n:
            raise ValueError('truncated protobuf message')
        yield fn, wt, v


def _packed_varints(v, wt):
    if wt == 0:
        return [v]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _sint64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf):
    dims, dtype, name, raw = [], 1, '', None
    floats, int64s, doubles = [], [], []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            dims += [_sint64(x) for x in _packed_varints(v, wt)]
        elif fn == 2:
            dtype = v
        elif fn == 4:
            floats.append(np.frombuffer(bytes(v), '<f4') if wt == 2 else np.frombuffer(v, '<f4'))
        elif fn == 7:
            int64s += [_sint64(x) for x in _packed_varints(v, wt)]
        elif fn == 8:
            name = bytes(v).decode('utf-8')
        elif fn == 9:
            raw = bytes(v)
        elif fn == 10:
            doubles.append(np.frombuffer(bytes(v), '<f8') if wt == 2 else np.frombuffer(v, '<f8'))
        elif fn == 13 or fn == 14:
            raise NotImplementedError(f'tensor {name!r}: external data is not supported')
    if dtype not in _DTYPES:
        raise NotImplementedError(f'tensor {name!r}: ONNX data_type {dtype} is not supported')
    dt = np.dtype(_DTYPES[dtype]).newbyteorder('<')
    if raw is not None:
        arr = np.frombuffer(raw, dt)
    elif floats:
        arr = np.concatenate(floats).astype(dt)
    elif doubles:
        arr = np.concatenate(doubles).astype(dt)
    else:
        arr = np.asarray(int64s, dtype=dt)
    return name, arr.reshape(dims).astype(_DTYPES[dtype])


def _attribute(buf):
    name, val = '', None
    ints = []
    for fn, wt, v in _fields(buf):
        if fn == 1:
            name = bytes(v).decode('utf-8')
        elif fn == 2:
            val = struct.unpack('<f', v)[0]
        elif fn == 3:
            val = _sint64(v)
        elif fn == 8:
            ints += [_sint64(x) for x in _packed_varints(v, wt)]
    return name, (ints if ints else val)


def _node(buf):
    n = {'input': [], 'output': [], 'name': '', 'op_type': '', 'attr': {}}
    for fn, wt, v in _fields(buf):
        if fn == 1:
            n['input'].append(bytes(v).decode('utf-8'))
        elif fn == 2:
            n['output'].append(bytes(v).decode('utf-8'))
        elif fn == 3:
            n['name'] = bytes(v).decode('utf-8')
        elif fn == 4:
            n['op_type'] = bytes(v).decode('utf-8')
        elif fn == 5:
            k, a = _attribute(v)
            n['attr'][k] = a
    return n


def read_graph(path):
    """-> (nodes [dict], initializers {name: ndarray}) of the model's main graph."""
    buf = memoryview(open(path, 'rb').read())
    graph = None
    for fn, wt, v in _fields(buf):
        if fn == 7 and wt == 2:
            graph = v
    if graph is None:
        raise ValueError(f'{path}: no GraphProto (field 7) -- not an ONNX ModelProto')
    nodes, inits = [], {}
    for fn, wt, v in _fields(graph):
        if fn == 1 and wt == 2:
            nodes.append(_node(v))
        elif fn == 5 and wt == 2:
            name, arr = _tensor(v)
            inits[name] = arr
    # Constant nodes carry tensors too (exporters put small ones there)
    for n in nodes:
        if n['op_type'] == 'Constant':
            pass                                     # (value attribute: not needed for the weights below)
    return nodes, inits


# ------------------------------------------------------------------------------ ResNet-101 of resnet.py
def resnet101_conv_names(m_channels=32, num_blocks=(3, 4, 23, 3)):
    """[(state_dict prefix of the conv, of its BatchNorm, (out, in, kh, kw))] in the execution order of resnet.py's forward."""
    out = [('conv1', 'bn1', (m_channels, 1, 3, 3))]
    inp = m_channels
    for li, (planes, nb) in enumerate(zip((m_channels, 2 * m_channels, 4 * m_channels, 8 * m_channels), num_blocks), 1):
        for bi in range(nb):
            p = f'layer{li}.{bi}'
            out.append((p + '.conv1', p + '.bn1', (planes, inp, 1, 1)))
            out.append((p + '.conv2', p + '.bn2', (planes, planes, 3, 3)))
            out.append((p + '.conv3', p + '.bn3', (4 * planes, planes, 1, 1)))
            if bi == 0:                              # stride != 1 or in_planes != 4 * planes: true for every stage's first block
                out.append((p + '.shortcut.0', p + '.shortcut.1', (4 * planes, inp, 1, 1)))
            inp = 4 * planes
    return out


_PASS_THROUGH = ('BatchNormalization', 'Relu', 'Identity', 'Cast')
# shape-only ops an exporter puts between the graph input and the stem convolution: resnet.py:116 does `x.unsqueeze_(1)` on the
# (1, feat, T) tensor the reference feeds (vbx_segmenter.py:265), so a torch export of `final.onnx` has an Unsqueeze there
_SHAPE_ONLY = ('Unsqueeze', 'Squeeze', 'Reshape', 'Transpose', 'Flatten')


def _order_convs_by_connectivity(path, nodes, convs, produced, consumers):
    """The Conv nodes in the order of `resnet101_conv_names()`, found by walking the GRAPH, not by trusting the node order: in
    the first block of a stage conv3 and the shortcut projection have the same weight shape (layer1.0: both (128, 32, 1, 1)), so
    an exporter / simplifier that lists them the other way round would swap them silently if only order + shape were used.
    Rules (resnet.py:60-75): a Bottleneck's conv1 and its shortcut projection read the SAME tensor (the block input); conv1 is
    the one whose output reaches another Conv (conv2) through BatchNormalization / Relu, the shortcut's reaches the Add; conv2
    reads conv1's output, conv3 conv2's; the Add both branches meet in produces the next block's input."""
    def source(t):                                    # the Conv / Add / graph input a tensor comes from (through BN / Relu / ...)
        seen = 0
        while True:
            n = produced.get(t)
            if n is None:
                return None
            if n['op_type'] in _SHAPE_ONLY and n['input']:
                # a shape-only op counts as transparent only on the way back to a GRAPH INPUT (the stem); anywhere else it would
                # be a topology this reader does not know
                u, hops = n['input'][0], 0
                while u in produced and produced[u]['op_type'] in _SHAPE_ONLY + ('Identity', 'Cast') and produced[u]['input'] and hops < 16:
                    u, hops = produced[u]['input'][0], hops + 1
                return None if u not in produced else id(n)
            if n['op_type'] not in _PASS_THROUGH or not n['input']:
                return id(n)
            t = n['input'][0]
            seen += 1
            if seen > 64:
                raise ValueError(f'{path}: pass-through chain too long behind {t!r}')

    by_source = {}
    for c in convs:
        by_source.setdefault(source(c['input'][0]), []).append(c)

    def next_join(c):                                 # the Add a conv's output reaches through BN / Relu (or None)
        t = c['output'][0]
        for _ in range(64):
            cs = consumers.get(t, [])
            adds = [n for n in cs if n['op_type'] == 'Add']
            if adds:
                return adds[0]
            nxt = [n for n in cs if n['op_type'] in _PASS_THROUGH]
            if not nxt:
                return None
            t = nxt[0]['output'][0]
        return None

    stem = by_source.get(None, [])
    if len(stem) != 1:
        raise ValueError(f'{path}: {len(stem)} Conv nodes read the graph input, expected the one 3x3 stem convolution')
    ordered = [stem[0]]
    x = id(stem[0])                                   # what the next block's convolutions must come from
    nblocks = sum((3, 4, 23, 3))
    for b in range(nblocks):
        heads = by_source.get(x, [])
        c1 = [c for c in heads if by_source.get(id(c))]              # its output feeds another Conv: conv1
        sc = [c for c in heads if not by_source.get(id(c))]          # ... feeds the Add: the shortcut projection
        if len(c1) != 1 or len(sc) > 1:
            raise ValueError(f'{path}: block {b}: {len(heads)} Conv nodes read the block input ({len(c1)} feeding a Conv); '
                             'this is not the Bottleneck of resnet.py')
        c2 = by_source[id(c1[0])]
        c3 = by_source.get(id(c2[0]), []) if len(c2) == 1 else []
        if len(c2) != 1 or len(c3) != 1:
            raise ValueError(f'{path}: block {b}: conv1 -> conv2 -> conv3 chain not found')
        ordered += [c1[0], c2[0], c3[0]] + sc
        join = next_join(c3[0])
        if join is None:
            raise ValueError(f'{path}: block {b}: conv3 does not reach a residual Add')
        if sc and next_join(sc[0]) is not join:
            raise ValueError(f'{path}: block {b}: the shortcut projection and conv3 do not meet in the same Add')
        x = id(join)
    if len(ordered) != len(convs) or len({id(c) for c in ordered}) != len(convs):
        raise ValueError(f'{path}: {len(convs)} Conv nodes but the Bottleneck walk found {len(ordered)}')
    return ordered


def _check_conv_geometry(path, convs, want):
    """strides / pads / group of every Conv against resnet.py:48-75,95-113 (stage strides 1, 2, 2, 2 on conv2 and on the
    shortcut projection; 3x3 padded by 1, 1x1 unpadded; no grouped convolution)."""
    stage_stride = {1: 1, 2: 2, 3: 2, 4: 2}
    for node, (cname, _, shape) in zip(convs, want):
        a = node['attr']
        k = shape[2]
        stride = 1
        if cname != 'conv1':
            li, bi, leaf = cname[5:].split('.', 2)
            if bi == '0' and (leaf == 'conv2' or leaf.startswith('shortcut')):
                stride = stage_stride[int(li)]
        st = a.get('strides') or [1, 1]
        pd = a.get('pads') or [0, 0, 0, 0]
        st = [st] * 2 if isinstance(st, int) else list(st)
        pd = [pd] * 4 if isinstance(pd, int) else list(pd)
        dl = a.get('dilations') or [1, 1]
        dl = [dl] * 2 if isinstance(dl, int) else list(dl)
        if st != [stride, stride] or pd != [k // 2] * 4 or (a.get('group') or 1) != 1 or dl != [1, 1]:
            raise ValueError(f'{path}: {cname}: strides {st} pads {pd} group {a.get("group")} dilations {dl}, resnet.py has strides '
                             f'{[stride, stride]} pads {[k // 2] * 4} group 1 dilations [1, 1]')


def load_resnet101_params(path):
    """`final.onnx` -> state_dict-like {name: float32 ndarray} (see the module docstring)."""
    nodes, inits = read_graph(path)
    produced = {o: n for n in nodes for o in n['output']}
    consumers = {}
    for n in nodes:
        for i in n['input']:
            consumers.setdefault(i, []).append(n)
    convs = [n for n in nodes if n['op_type'] == 'Conv']
    want = resnet101_conv_names()
    if len(convs) != len(want):
        raise ValueError(f'{path}: {len(convs)} Conv nodes, the ResNet-101 of resnet.py has {len(want)}')
    convs = _order_convs_by_connectivity(path, nodes, convs, produced, consumers)
    _check_conv_geometry(path, convs, want)
    params = {}

    def tensor_of(name):
        if name in inits:
            return inits[name]
        n = produced.get(name)
        if n is not None and n['op_type'] in ('Identity', 'Cast') and n['input']:
            return tensor_of(n['input'][0])
        raise ValueError(f'{path}: tensor {name!r} is not an initializer')

    for node, (cname, bname, shape) in zip(convs, want):
        W = np.asarray(tensor_of(node['input'][1]), np.float32)
        if tuple(W.shape) != shape:
            raise ValueError(f'{path}: Conv #{convs.index(node)} has weights {tuple(W.shape)}, {cname} expects {shape}')
        params[cname + '.weight'] = W
        nxt = consumers.get(node['output'][0], [])
        bn = next((c for c in nxt if c['op_type'] == 'BatchNormalization'), None)
        if bn is not None:
            if len(node['input']) > 2:
                raise NotImplementedError(f'{path}: {cname} has a bias AND a BatchNormalization')
            g, b, mu, var = (np.asarray(tensor_of(x), np.float32) for x in bn['input'][1:5])
            eps = bn['attr'].get('epsilon', 1e-5)
            if abs(eps - 1e-5) > 1e-12:
                # compile_resnet101 folds with eps = 1e-5 (torch's default): re-express a different epsilon in the variance
                var = (var.astype(np.float64) + eps - 1e-5).astype(np.float32)
            params[bname + '.weight'], params[bname + '.bias'] = g, b
            params[bname + '.running_mean'], params[bname + '.running_var'] = mu, var
        else:
            if len(node['input']) < 3:
                raise ValueError(f'{path}: {cname} has neither a bias nor a BatchNormalization behind it')
            params[cname + '.bias'] = np.asarray(tensor_of(node['input'][2]), np.float32)
    # the embedding layer: Gemm (weights (256, 16384) with transB = 1, or transposed) or MatMul + Add
    emb = [n for n in nodes if n['op_type'] in ('Gemm', 'MatMul')]
    if not emb:
        raise ValueError(f'{path}: no Gemm / MatMul node (the embedding layer)')
    node = emb[-1]
    W = np.asarray(tensor_of(node['input'][1]), np.float32)
    if node['op_type'] == 'Gemm':
        if not node['attr'].get('transB', 0):
            W = W.T
        b = np.asarray(tensor_of(node['input'][2]), np.float32) if len(node['input']) > 2 else np.zeros(W.shape[0], np.float32)
        b = b * np.float32(node['attr'].get('beta', 1.0))
        W = W * np.float32(node['attr'].get('alpha', 1.0))
    else:
        W = W.T
        add = next((c for c in consumers.get(node['output'][0], []) if c['op_type'] == 'Add'), None)
        b = np.zeros(W.shape[0], np.float32)
        if add is not None:
            other = [x for x in add['input'] if x != node['output'][0]][0]
            b = np.asarray(tensor_of(other), np.float32)
    if W.ndim != 2 or W.shape[1] != 16384:
        raise ValueError(f'{path}: embedding weights {W.shape}, expected (embed_dim, 16384)')
    params['embedding.weight'] = np.ascontiguousarray(W)
    params['embedding.bias'] = b.reshape(-1)
    return params
