"""CPU: the package's own ONNX reader (inaspeechsegmenter_amd/onnx_reader.py) on hand-built files.

`final.onnx` -- what the reference's live x-vector backend loads (vbx_segmenter.py:249-266, remote_utils.py:13) -- is an
un-vendored release asset, so the reader is pinned on files written here, field by field, by a tiny protobuf ENCODER (the
`onnx` package is not installed): the seeded ResNet-101 of resnet.py in the two styles a torch -> ONNX export produces
(BatchNormalization nodes kept with the original initializer names; BatchNorm folded into Conv with anonymous
`onnx::Conv_N` names), float tensors as raw_data and as packed float_data, Gemm with transB and MatMul + Add.  The
parameters read back must reproduce the oracle's embedding of the original parameters."""
import struct

import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, onnx_reader as OR
from oracle import vbx as ovbx


# ------------------------------------------------------------------------------ a minimal protobuf encoder
def _vi(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _ld(fn, payload):
    return _vi(fn << 3 | 2) + _vi(len(payload)) + payload


def _str(fn, s):
    return _ld(fn, s.encode())


def _int(fn, v):
    return _vi(fn << 3 | 0) + _vi(v)


def tensor(name, arr, style):
    arr = np.ascontiguousarray(arr, np.float32)
    msg = b''.join(_int(1, d) for d in arr.shape) if style != 'packed_dims' else _ld(1, b''.join(_vi(d) for d in arr.shape))
    msg += _int(2, 1) + _str(8, name)
    msg += _ld(9, arr.tobytes()) if style != 'float_data' else _ld(4, arr.tobytes())
    return msg


def attr_i(name, v):
    return _str(1, name) + _int(3, v) + _int(20, 2)


def attr_f(name, v):
    return _str(1, name) + _vi(2 << 3 | 5) + struct.pack('<f', v) + _int(20, 1)


def attr_ints(name, vs):
    return _str(1, name) + b''.join(_int(8, v) for v in vs) + _int(20, 7)


def node(op, inputs, outputs, attrs=()):
    return b''.join(_str(1, i) for i in inputs) + b''.join(_str(2, o) for o in outputs) + _str(4, op) + b''.join(_ld(5, a) for a in attrs)


def write_resnet_onnx(path, params, folded, tensor_style='raw', embed='gemm', stem='direct'):
    """Serialise the ResNet-101 of resnet.py as an ONNX ModelProto: node order = torch's trace order."""
    nodes, inits = [], []
    counter = [0]

    def fresh(prefix):
        counter[0] += 1
        return f'{prefix}_{counter[0]}'

    def conv(x, cname, bname, stride, pad, k):
        W = params[cname + '.weight']
        if folded:
            sc = params[bname + '.weight'].astype(np.float64) / np.sqrt(params[bname + '.running_var'].astype(np.float64) + 1e-5)
            Wf = (W.astype(np.float64) * sc[:, None, None, None]).astype(np.float32)
            bf = (params[bname + '.bias'] - params[bname + '.running_mean'] * sc).astype(np.float32)
            wn, bn_ = fresh('onnx::Conv'), fresh('onnx::Conv')
            inits.extend([tensor(wn, Wf, tensor_style), tensor(bn_, bf, tensor_style)])
            y = fresh('conv_out')
            nodes.append(node('Conv', [x, wn, bn_], [y], [attr_ints('kernel_shape', [k, k]), attr_ints('strides', [stride, stride]),
                                                         attr_ints('pads', [pad] * 4)]))
            return y
        inits.append(tensor(cname + '.weight', W, tensor_style))
        y = fresh('conv_out')
        nodes.append(node('Conv', [x, cname + '.weight'], [y], [attr_ints('kernel_shape', [k, k]), attr_ints('strides', [stride, stride]),
                                                               attr_ints('pads', [pad] * 4)]))
        for suf in ('weight', 'bias', 'running_mean', 'running_var'):
            inits.append(tensor(f'{bname}.{suf}', params[f'{bname}.{suf}'], tensor_style))
        z = fresh('bn_out')
        nodes.append(node('BatchNormalization', [y] + [f'{bname}.{s}' for s in ('weight', 'bias', 'running_mean', 'running_var')], [z],
                          [attr_f('epsilon', 1e-5), attr_f('momentum', 0.9)]))
        return z

    def relu(x):
        y = fresh('relu')
        nodes.append(node('Relu', [x], [y]))
        return y

    x0 = 'input'
    if stem == 'unsqueeze':
        # what torch.onnx.export writes for resnet.py:116 (`x.unsqueeze_(1)` on the 3-D (1, feat, T) input of vbx_segmenter.py:265):
        # an Unsqueeze (axes as a second input, opset >= 13) -- here behind a Cast, as some exporters add one
        inits.append(tensor('onnx::Unsqueeze_axes', np.array([1], np.int64).astype(np.float32), tensor_style))
        nodes.append(node('Cast', ['input'], ['input_f32']))
        nodes.append(node('Unsqueeze', ['input_f32', 'onnx::Unsqueeze_axes'], ['input_4d']))
        x0 = 'input_4d'
    x = relu(conv(x0, 'conv1', 'bn1', 1, 1, 3))
    for li, (planes, nb, stride) in enumerate(zip((32, 64, 128, 256), (3, 4, 23, 3), (1, 2, 2, 2)), 1):
        for bi in range(nb):
            p = f'layer{li}.{bi}'
            s = stride if bi == 0 else 1
            o = relu(conv(x, p + '.conv1', p + '.bn1', 1, 0, 1))
            o = relu(conv(o, p + '.conv2', p + '.bn2', s, 1, 3))
            o = conv(o, p + '.conv3', p + '.bn3', 1, 0, 1)
            sc = conv(x, p + '.shortcut.0', p + '.shortcut.1', s, 0, 1) if bi == 0 else x
            y = fresh('add')
            nodes.append(node('Add', [o, sc], [y]))
            x = relu(y)
    nodes.append(node('ReduceMean', [x], ['pooled'], [attr_ints('axes', [-1])]))       # (statistics pooling, abbreviated: not read)
    if embed == 'gemm':
        inits.extend([tensor('embedding.weight', params['embedding.weight'], tensor_style),
                      tensor('embedding.bias', params['embedding.bias'], tensor_style)])
        nodes.append(node('Gemm', ['pooled', 'embedding.weight', 'embedding.bias'], ['output'], [attr_f('alpha', 1.0), attr_f('beta', 1.0), attr_i('transB', 1)]))
    else:
        inits.extend([tensor('onnx::MatMul_9', params['embedding.weight'].T.copy(), tensor_style),
                      tensor('embedding.bias', params['embedding.bias'], tensor_style)])
        nodes.append(node('MatMul', ['pooled', 'onnx::MatMul_9'], ['mm']))
        nodes.append(node('Add', ['embedding.bias', 'mm'], ['output']))
    graph = b''.join(_ld(1, n) for n in nodes) + _str(2, 'torch_jit') + b''.join(_ld(5, t) for t in inits)
    model = _int(1, 8) + _str(2, 'pytorch') + _ld(7, graph) + _ld(8, _str(1, '') + _int(2, 14))
    open(path, 'wb').write(model)


@pytest.mark.parametrize('folded,style,embed,stem', [(False, 'raw', 'gemm', 'direct'), (True, 'float_data', 'matmul', 'direct'),
                                                     (True, 'packed_dims', 'gemm', 'direct'), (True, 'raw', 'gemm', 'unsqueeze'),
                                                     (False, 'raw', 'matmul', 'unsqueeze')])
def test_reader_reproduces_the_network(tmp_path, folded, style, embed, stem):
    params = KM.synthetic_resnet101(3)
    path = str(tmp_path / 'final.onnx')
    write_resnet_onnx(path, params, folded, style, embed, stem)
    got = OR.load_resnet101_params(path)
    if not folded:                                   # kept BatchNormalization: the original tensors, bit for bit
        assert set(got) == set(params)
        for k in params:
            assert np.array_equal(got[k], params[k]), k
    else:
        assert 'bn1.weight' not in got and 'conv1.bias' in got
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (2, 64, 48)).astype(np.float32)
    want = ovbx.resnet101_forward(params, x)
    have = ovbx.resnet101_forward(_unfold(got), x)
    assert np.abs(have - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    # and the lowering the device runs accepts both forms
    net = KM.compile_resnet101(got, frames=48)
    ref = KM.compile_resnet101(params, frames=48)
    assert net.prog.shape == ref.prog.shape and np.allclose(net.blob, ref.blob, atol=2e-6)


def _unfold(p):
    """oracle-side view of a folded export: identity BatchNorms around the biased convolutions"""
    if 'bn1.weight' in p:
        return p
    q = dict(p)
    for cname, bname, shape in OR.resnet101_conv_names():
        c = shape[0]
        q[bname + '.weight'] = np.ones(c, np.float32)
        q[bname + '.bias'] = p[cname + '.bias']
        q[bname + '.running_mean'] = np.zeros(c, np.float32)
        q[bname + '.running_var'] = np.full(c, 1.0 - 1e-5, np.float32)
    return q


def test_reader_rejects_other_graphs(tmp_path):
    params = KM.synthetic_resnet101(1)
    path = str(tmp_path / 'x.onnx')
    bad = dict(params)
    bad['layer2.0.conv2.weight'] = np.zeros((64, 64, 1, 1), np.float32)            # wrong kernel size for that position
    write_resnet_onnx(path, bad, False)
    with pytest.raises(ValueError):
        OR.load_resnet101_params(path)
    open(path, 'wb').write(b'\x08\x08')
    with pytest.raises(ValueError):
        OR.load_resnet101_params(path)


def test_reordered_export_is_matched_by_connectivity_not_by_order(tmp_path, monkeypatch):
    """layer1.0's conv3 and shortcut projection are both (128, 32, 1, 1): an exporter that lists the shortcut FIRST must still
    give each its own weights (the reader walks the graph); wrong strides / pads are refused."""
    params = KM.synthetic_resnet101(5)
    path = str(tmp_path / 'final.onnx')
    write_resnet_onnx(path, params, False)
    nodes, inits = OR.read_graph(path)
    # permute the node list: every shortcut Conv (+ its BatchNormalization) moves in front of its block's conv1, and the whole
    # list of the first block is reversed on top (readers must not depend on topological order either)
    conv_idx = [i for i, n in enumerate(nodes) if n['op_type'] == 'Conv']
    order = list(range(len(nodes)))
    sc_conv = conv_idx[4]                              # stem, l1.0.conv1, conv2, conv3, shortcut
    blk = order[conv_idx[1]:sc_conv + 2]
    order[conv_idx[1]:sc_conv + 2] = blk[-2:] + blk[:-2]
    shuffled = [nodes[i] for i in order]
    monkeypatch.setattr(OR, 'read_graph', lambda p: (shuffled, inits))
    got = OR.load_resnet101_params(path)
    for k in ('layer1.0.conv3.weight', 'layer1.0.shortcut.0.weight', 'layer1.0.bn3.bias', 'layer1.0.shortcut.1.bias'):
        assert np.array_equal(got[k], params[k]), k
    assert not np.array_equal(params['layer1.0.conv3.weight'], params['layer1.0.shortcut.0.weight'])
    # geometry: a stride-1 stage-2 shortcut is not resnet.py's network
    for n in shuffled:
        n['attr'] = dict(n['attr'])
    stage2_sc = [n for n in nodes if n['op_type'] == 'Conv'][1 + 3 * 3 + 1 + 3]       # layer2.0.shortcut.0
    stage2_sc['attr']['strides'] = [1, 1]
    with pytest.raises(ValueError, match='strides'):
        OR.load_resnet101_params(path)
