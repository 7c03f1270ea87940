"""CPU: host-side product code (no device calls): C-ABI exports, compiled Viterbi, tables, WAV
reader, exporters, window arithmetic, Keras lowering, sharding."""
import ctypes
import io
import json
import os
import re
import sys
import warnings

import numpy as np
import pytest

from inaspeechsegmenter_amd import _native, tables, segmenter as S, keras_model as KM
from inaspeechsegmenter_amd import export_funcs, io as iss_io
from oracle import segment as oseg, sidekit as osk, viterbi as ovit, vbx as ovbx, keras_cnn as ocnn
from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'iss.h')).read()
    declared = set(re.findall(r'\b(iss_[a-z0-9_]+)\s*\(', hdr))
    L = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(_native.lib()._iss_symbols)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'inaspeechsegmenter_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_native.NativeError):
        _native.Context(0)


def test_compiled_viterbi_golden_and_oracle():
    g = np.load(os.path.join(GOLDEN, 'viterbi_cases.npz'))
    for i in range(6):
        assert np.array_equal(_native.viterbi(g[f'em{i}'], g[f'tr{i}']), g[f'st{i}'].astype(np.int32))
    rng = np.random.default_rng(3)
    for K in (2, 3, 5):
        em = np.log(rng.dirichlet(np.ones(K), 3000))
        em[rng.random(3000) < 0.1] = np.log(0.5)
        em[5] = -np.inf
        tr = S.diag_trans_exp(3, K)
        assert np.array_equal(_native.viterbi(em, tr), ovit.viterbi_decoding(em, tr).astype(np.int32))
        assert np.array_equal(_native.viterbi(em.astype(np.float32), tr),
                              ovit.viterbi_decoding(em.astype(np.float32), tr).astype(np.int32))


def test_energy_activity_golden():
    f = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    g = np.load(os.path.join(GOLDEN, 'viterbi_cases.npz'))
    assert np.array_equal(S._energy_activity(f['musanmix_loge'], 0.03), g['energy_states_musanmix'])
    assert not S._energy_activity(f['silence_loge'], 0.03).any()
    assert np.array_equal(S.pred2logemission([0, 1, 1]), ovit.pred2logemission([0, 1, 1]))
    assert np.array_equal(S.log_trans_exp(150, cost0=-5), ovit.log_trans_exp(150, cost0=-5))
    assert np.array_equal(S.diag_trans_exp(80, 3), ovit.diag_trans_exp(80, 3))


def test_tables_match_reference():
    assert np.array_equal(tables.sidekit_melbank(), np.load(os.path.join(GOLDEN, 'sidekit_melbank.npy')))
    assert np.array_equal(tables.vbx_melbank(), ovbx.mel_bank())
    assert np.array_equal(tables.vbx_window(), ovbx.povey_window())


def test_window_rows_equal_get_patches():
    rng = np.random.default_rng(0)
    for T in (68, 69, 70, 71, 131, 298):
        m = rng.normal(0, 1, (T, 24)).astype(np.float32)
        patches, _ = oseg.get_patches(m, 68, 2)
        rows = S._window_rows(T)
        assert len(rows) == len(patches) == -(-T // 2)
        for i in range(len(rows)):
            w = m[rows[i]:rows[i] + 68]
            assert np.allclose((w - w.mean()) / w.std(), patches[i], atol=2e-5)
    assert len(S._window_rows(68, difflen=2)) == 33          # 66-frame media -> 33 slots (0021.mp3)


def test_binidx2seglist():
    for seq in ([0], [1, 1], [0, 0, 1, 1, 1, 0], list('ffbbbv')):
        assert S._binidx2seglist(np.array(seq)) == oseg.binidx2seglist(seq)


def test_wav_reader(tmp_path):
    pcm = iss_io.decode_pcm(os.path.join(GOLDEN, 'musanmix.wav'), ffmpeg=None)      # has a LIST chunk
    assert pcm.dtype == np.int16 and pcm.shape == (1192367,)
    f = iss_io.decode_pcm(os.path.join(GOLDEN, 'lamartine.wav'), ffmpeg=None)        # IEEE float + extra chunks
    assert f.dtype == np.float32 and f.shape == (234282,)
    import scipy.io.wavfile as wf
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        sr, ref = wf.read(os.path.join(GOLDEN, 'lamartine.wav'))
    assert np.array_equal(ref, f)
    s64 = iss_io.media2sig16kmono(os.path.join(GOLDEN, 'musanmix.wav'), ffmpeg=None)
    assert s64.dtype == np.float64 and np.array_equal(s64, pcm / 32768.0)
    with pytest.raises(NotImplementedError):
        iss_io.media2sig16kmono('x.wav', start_sec=1.0, ffmpeg=None)
    with pytest.raises(NotImplementedError):
        iss_io.media2sig16kmono('http://a/b.wav', ffmpeg=None)
    # 8 kHz file is refused like the reference (io.py:53-55)
    import struct
    p = tmp_path / 'a.wav'
    d = np.zeros(100, np.int16).tobytes()
    p.write_bytes(b'RIFF' + struct.pack('<I', 36 + len(d)) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 8000, 16000, 2, 16)
                  + b'data' + struct.pack('<I', len(d)) + d)
    with pytest.raises(AssertionError):
        iss_io.decode_pcm(str(p), ffmpeg=None)


def test_exporters_byte_identical_to_reference_goldens(tmp_path):
    rows = [l.rstrip('\n').split('\t') for l in open(os.path.join(GOLDEN, 'musanmix-smn-gender.csv'))][1:]
    lseg = [(r[0], float(r[1]), float(r[2])) for r in rows]
    out = tmp_path / 'a.csv'
    export_funcs.seg2csv(lseg, str(out))
    assert out.read_bytes() == open(os.path.join(GOLDEN, 'musanmix-smn-gender.csv'), 'rb').read()
    tg = tmp_path / 'a.TextGrid'
    export_funcs.seg2textgrid(lseg, str(tg))
    assert tg.read_bytes() == open(os.path.join(GOLDEN, 'musanmix-smn-gender.TextGrid'), 'rb').read()
    # and against pandas itself on awkward floats
    import pandas as pd
    rng = np.random.default_rng(1)
    lseg = [('speech', float(a) * .02, float(b) * .02 + 3.0) for a, b in rng.integers(0, 10 ** 6, (200, 2))]
    s = io.StringIO()
    pd.DataFrame.from_records(lseg, columns=['labels', 'start', 'stop']).to_csv(s, sep='\t', index=False)
    t = io.StringIO()
    export_funcs.seg2csv(lseg, t)
    assert s.getvalue() == t.getvalue()


def _keras_cfg_and_weights(rng):
    cfg = {'class_name': 'Sequential', 'config': {'name': 'm', 'layers': [
        {'class_name': 'Conv2D', 'config': {'name': 'c1', 'batch_input_shape': [None, 68, 21, 1], 'filters': 8,
                                            'kernel_size': [3, 3], 'strides': [1, 1], 'padding': 'same',
                                            'activation': 'linear', 'use_bias': True}},
        {'class_name': 'BatchNormalization', 'config': {'name': 'b1', 'axis': [3], 'epsilon': 0.001}},
        {'class_name': 'Activation', 'config': {'name': 'a1', 'activation': 'relu'}},
        {'class_name': 'MaxPooling2D', 'config': {'name': 'p1', 'pool_size': [2, 2], 'strides': None, 'padding': 'valid'}},
        {'class_name': 'Dropout', 'config': {'name': 'd', 'rate': 0.5}},
        {'class_name': 'Flatten', 'config': {'name': 'f'}},
        {'class_name': 'Dense', 'config': {'name': 'fc', 'units': 3, 'activation': 'softmax', 'use_bias': True}}]}}
    w = {'c1': {'kernel': rng.normal(0, 1, (3, 3, 1, 8)), 'bias': rng.normal(0, 1, 8)},
         'b1': {'gamma': rng.uniform(.5, 1.5, 8), 'beta': rng.normal(0, 1, 8), 'moving_mean': rng.normal(0, 1, 8),
                'moving_variance': rng.uniform(.5, 1.5, 8)},
         'fc': {'kernel': rng.normal(0, .05, (34 * 10 * 8, 3)), 'bias': rng.normal(0, 1, 3)}}
    return cfg, w


def test_keras_config_parse_and_lowering(tmp_path):
    rng = np.random.default_rng(0)
    cfg, w = _keras_cfg_and_weights(rng)
    layers, shp = KM.layers_from_keras_config(cfg, w)
    assert shp == (68, 21, 1) and [l['type'] for l in layers] == ['conv2d', 'batchnorm', 'activation', 'maxpool',
                                                                  'dropout', 'flatten', 'dense']
    c = KM.compile_layers(layers, shp, fuse_pool=False)
    ops = list(c.prog[:, _native.C_OP])
    assert ops == [_native.OP_CONV, _native.OP_POOL, _native.OP_CONV, _native.OP_SOFTMAX]   # BN+relu fused
    assert c.prog[0, _native.C_INMODE] == 1 and c.prog[0, _native.C_ACT] == 1 and c.out_dim == 3
    assert c.flops_per_sample == ocnn.flops_per_sample(layers, shp)
    cf = KM.compile_layers(layers, shp)                                                    # 2x2 max-pool -> conv epilogue
    assert list(cf.prog[:, _native.C_OP]) == [_native.OP_CONV, _native.OP_CONV, _native.OP_SOFTMAX]
    assert tuple(cf.prog[0, [_native.C_FPOOLH, _native.C_FPOOLW, _native.C_POOLKIND]]) == (2, 2, 0)
    assert tuple(cf.prog[0, [_native.C_HO, _native.C_WO]]) == (68, 21) and cf.buf_elems[cf.prog[0, _native.C_OUT]] == 34 * 10 * 8
    assert cf.flops_per_sample == c.flops_per_sample - 2 * 9 * 8 * (68 * 21 - 68 * 20)       # the odd 21st column is never computed
    assert cf.prog[0, _native.C_WOFF] % 8 == 0 and cf.prog[1, _native.C_WOFF] % 8 == 0
    # flat .npz round trip (what scripts/convert_keras_hdf5.py writes)
    flat = {'model_config': np.array(json.dumps(cfg))}
    for ln, d in w.items():
        for wn, a in d.items():
            flat[f'{ln}/{ln}/{wn}:0'] = a
    np.savez(tmp_path / 'm.npz', **flat)
    layers2, shp2 = KM.load_model_file(str(tmp_path / 'm.npz'))
    assert shp2 == shp and np.array_equal(layers2[0]['W'], layers[0]['W'])
    with pytest.raises(NotImplementedError):
        KM.layers_from_keras_config({'class_name': 'Sequential', 'config': {'layers': [
            {'class_name': 'LSTM', 'config': {'name': 'l', 'batch_input_shape': [None, 68, 21, 1]}}]}}, {})


def test_oracle_cnn_two_implementations_agree():
    rng = np.random.default_rng(2)
    cfg, w = _keras_cfg_and_weights(rng)
    layers, shp = KM.layers_from_keras_config(cfg, w)
    x = rng.normal(0, 1, (3, 68, 21, 1)).astype(np.float32)
    a, b = ocnn.forward(layers, x), ocnn.forward_naive(layers, x)
    assert np.abs(a - b).max() < 1e-5 and np.allclose(a.sum(1), 1, atol=1e-5)
    layers, shp = KM.synthetic_ina_like(21, 3, seed=1)
    x = rng.normal(0, 1, (2, 68, 21, 1)).astype(np.float32)
    assert np.abs(ocnn.forward(layers, x) - ocnn.forward_naive(layers, x)).max() < 1e-4


def test_resnet_lowering_shapes():
    params = ovbx.resnet101_random_params(0)
    c = KM.compile_resnet101(params)
    assert c.out_dim == 256 and c.in_shape == (64, 144, 1)
    assert abs(c.flops_per_sample / 2 - 5.65e9) < 0.05e9          # 5.65 GMAC / window
    assert sum(a.size for a in params.values() if a.ndim > 1) < c.blob.size


def test_model_locator_message():
    with pytest.raises(FileNotFoundError) as e:
        S.locate_model('keras_speech_music_noise_cnn.hdf5')
    assert 'releases/download/models' in str(e.value)


def test_hdf5_converter_roundtrip(tmp_path):
    """tools/convert_keras_hdf5.py on a file laid out like Keras' HDF5 (model_config attribute,
    model_weights/<layer>/<layer>/<weight>:0 datasets + weight_names attributes); needs an interpreter
    with h5py (the image has one under /opt/conda)."""
    import shutil
    import subprocess
    py = next((p for p in ('/opt/conda/bin/python3.9', shutil.which('python3') or '') if p and subprocess.run(
        [p, '-c', 'import h5py'], capture_output=True).returncode == 0), None)
    if py is None:
        pytest.skip('no interpreter with h5py')
    rng = np.random.default_rng(4)
    cfg, w = _keras_cfg_and_weights(rng)
    np.savez(tmp_path / 'w.npz', **{f'{ln}|{wn}': a.astype(np.float32) for ln, d in w.items() for wn, a in d.items()})
    (tmp_path / 'cfg.json').write_text(json.dumps(cfg))
    mk = ("import h5py, json, numpy as np, sys\n"
          "z = np.load(sys.argv[1]); cfg = open(sys.argv[2]).read()\n"
          "f = h5py.File(sys.argv[3], 'w'); f.attrs['model_config'] = cfg.encode('utf-8'); g = f.create_group('model_weights')\n"
          "names = {}\n"
          "for k in z.files:\n"
          "    ln, wn = k.split('|'); names.setdefault(ln, []).append(wn)\n"
          "    g.require_group(ln).require_group(ln).create_dataset(wn + ':0', data=z[k])\n"
          "for ln, ws in names.items():\n"
          "    g[ln].attrs['weight_names'] = [(ln + '/' + wn + ':0').encode() for wn in ws]\n"
          "f.close()\n")
    subprocess.run([py, '-c', mk, str(tmp_path / 'w.npz'), str(tmp_path / 'cfg.json'), str(tmp_path / 'm.hdf5')], check=True)
    tool = os.path.join(os.path.dirname(GOLDEN), '..', 'tools', 'convert_keras_hdf5.py')
    subprocess.run([py, tool, str(tmp_path / 'm.hdf5')], check=True)
    layers, shp = KM.load_model_file(str(tmp_path / 'm.npz'))
    ref_layers, _ = KM.layers_from_keras_config(cfg, w)
    assert shp == (68, 21, 1) and len(layers) == len(ref_layers)
    for a, b in zip(layers, ref_layers):
        for k in ('W', 'b', 'gamma', 'beta', 'mean', 'var'):
            if k in b and b[k] is not None:
                assert np.array_equal(a[k], np.asarray(b[k], np.float32)), (a['type'], k)


def test_fast_div_magic_numbers():
    """The footprint kernel decomposes GEMM rows with host-precomputed reciprocals (conv_common.h set_fast_div / fast_div:
    q = mulhi(n, M) >> s with M = floor(2^(31+l) / d) + 1, s = l - 1, l = ceil(log2 d)).  Exact for 0 <= n < 2^31
    (Granlund-Montgomery, N = 31); this restates the formula and checks it, including the divisors the nets produce."""
    import random

    def magic(d):
        if d <= 1:
            return 0, 0
        l = 0
        while (1 << l) < d:
            l += 1
        return ((1 << (31 + l)) // d) + 1, l - 1

    rnd = random.Random(5)
    for d in list(range(1, 130)) + [915, 1098, 1105, 1300, 65535, 65536, 65537, 999983, 2 ** 30 - 1, 2 ** 30, 2 ** 31 - 1]:
        m, sh = magic(d)
        assert m < 2 ** 32
        top = (2 ** 31 - 1) // d * d
        for n in [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, top - 1, top, 2 ** 31 - 1] + [rnd.randrange(2 ** 31) for _ in range(300)]:
            if 0 <= n < 2 ** 31:
                q = n if m == 0 else ((n * m) >> 32) >> sh
                assert q == n // d, (d, n)


class _FakeCtx:
    """Stands in for the device context in CPU tests of the host mirror: holds the 'resident' mel spectrogram and
    answers cnn_probs(net, win_rows) with tests/golden/fake_predict.py on windows it z-normalises itself."""

    def __init__(self, predicts, nmels):
        self.predicts, self.nmels, self.mspec = predicts, nmels, None

    def set_mspec(self, m):
        self.mspec = np.asarray(m, dtype=np.float32)

    def cnn_probs(self, net_id, win_rows):
        h = self.nmels[net_id]
        pats = np.stack([self.mspec[r:r + 68, :h] for r in win_rows]).reshape(len(win_rows), -1)
        with np.errstate(divide='ignore', invalid='ignore'):
            pats = (pats - np.mean(pats, axis=1).reshape(-1, 1)) / np.std(pats, axis=1).reshape(-1, 1)
        fin = np.all(np.isfinite(pats), axis=1)
        p = self.predicts[net_id](pats.reshape(-1, 68, h, 1))
        p[~fin] = 0.5                                             # what iss_cnn_probs does (segmenter.py:175)
        return p, fin


@pytest.mark.parametrize('engine,nvad', [('smn', 3), ('sm', 2)])
def test_host_mirror_bookkeeping_equals_reference_lines(engine, nvad):
    """Segmenter.segment_feats / DnnSegmenter.__call__ / _window_rows / _energy_activity / compiled Viterbi of the PRODUCT,
    with the device replaced by a fake context, reproduce what the reference's own lines (segmenter.py:69-108,135-179,
    250-276, executed by tests/golden/ref_segmenter_pin.py) returned for the same inputs and the same stand-in predictor."""
    import sys
    sys.path.insert(0, GOLDEN)
    from fake_predict import make_predict
    pin = np.load(os.path.join(GOLDEN, 'segmenter_pin.npz'))
    feats = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    fake = _FakeCtx({0: make_predict(nvad, 1), 1: make_predict(2, 2)}, {0: 21, 1: 24})
    seg = object.__new__(S.Segmenter)
    seg.energy_ratio, seg.detect_gender, seg.ctx = 0.03, True, fake
    seg.vad = object.__new__(S.SpeechMusicNoise if engine == 'smn' else S.SpeechMusic)
    seg.gender = object.__new__(S.Gender)
    seg.vad.ctx = seg.gender.ctx = fake
    for tag in ('musanmix', 'silence', 'synth', 'short'):
        if f'{engine}_{tag}_labels' not in pin.files:
            continue
        if tag == 'short':
            mspec, difflen = pin['short_padded_mspec'], int(pin['short_difflen'])
        else:
            mspec, difflen = feats[tag + '_mspec'], 0
        got = seg.segment_feats(mspec, feats[tag + '_loge'], difflen, 0)
        assert [g[0] for g in got] == list(pin[f'{engine}_{tag}_labels']), tag
        assert np.array_equal(np.array([[a, b] for _, a, b in got], dtype=np.float64).reshape(-1, 2),
                              pin[f'{engine}_{tag}_bounds'].reshape(-1, 2)), tag


def test_ffmpeg_decode_path_with_a_shim(tmp_path):
    """io.py:56-79 through a stand-in `ffmpeg` executable (the image has none): the command line must be the reference's
    (io.py:61-68: -i <media> -f wav -acodec pcm_s16le -ar 16000 -ac 1 [-ss %f] [-to %f] pipe:1), the WAV arrives on a pipe
    (RIFF / data sizes unknown = 0xFFFFFFFF, an extra LIST chunk in front of the data) and a non-zero exit raises with stderr."""
    import stat
    import struct
    import subprocess
    pcm = (np.arange(48000) % 2000 - 1000).astype('<i2')
    src = tmp_path / 'in.mp3'
    src.write_bytes(pcm.tobytes())                          # the shim "decodes" raw PCM16
    shim = tmp_path / 'ffmpeg'
    shim.write_text(f'''#!{sys.executable}
import struct, sys
a = sys.argv[1:]
open({str(tmp_path / "argv.txt")!r}, 'w').write('\\n'.join(a))
if 'broken' in a[1]:
    sys.stderr.write('Invalid data found when processing input'); sys.exit(1)
pcm = open(a[1], 'rb').read()
ss = float(a[a.index('-ss') + 1]) if '-ss' in a else 0.0
to = float(a[a.index('-to') + 1]) if '-to' in a else len(pcm) / 32000.0
pcm = pcm[int(ss * 16000) * 2:int(to * 16000) * 2]
out = sys.stdout.buffer
out.write(b'RIFF' + struct.pack('<I', 0xFFFFFFFF) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 16000, 32000, 2, 16))
out.write(b'LIST' + struct.pack('<I', 26) + b'INFOISFT' + struct.pack('<I', 14) + b'Lavf58.76.100\\0')
out.write(b'data' + struct.pack('<I', 0xFFFFFFFF) + pcm)
''')
    shim.chmod(shim.stat().st_mode | stat.S_IEXEC)
    import sys as _sys  # noqa: F401
    a = iss_io.decode_pcm(str(src), None, None, str(shim))
    assert a.dtype == np.int16 and np.array_equal(a, pcm)
    assert (tmp_path / 'argv.txt').read_text().split('\n') == ['-i', str(src), '-f', 'wav', '-acodec', 'pcm_s16le', '-ar', '16000',
                                                                '-ac', '1', 'pipe:1']
    b = iss_io.decode_pcm(str(src), 0.5, 2.25, str(shim))
    assert np.array_equal(b, pcm[8000:36000])
    assert (tmp_path / 'argv.txt').read_text().split('\n') == ['-i', str(src), '-f', 'wav', '-acodec', 'pcm_s16le', '-ar', '16000',
                                                                '-ac', '1', '-ss', '0.500000', '-to', '2.250000', 'pipe:1']
    f = iss_io.media2sig16kmono(str(src), 0.5, None, str(shim), 'float32')      # reference signature / float result
    assert f.dtype == np.float32 and np.array_equal(f, (pcm[8000:] / 32768.0).astype(np.float32))
    bad = tmp_path / 'broken.mp3'
    bad.write_bytes(b'xx')
    with pytest.raises(Exception, match='Invalid data'):
        iss_io.decode_pcm(str(bad), None, None, str(shim))


class _FakeDevice(_FakeCtx):
    """_FakeCtx + the signal half of the context API: `sidekit()` runs the oracle's feature path on whatever PCM16 was
    handed over, the way the device does (frame t = samples [160 t, 160 t + 400) of the buffer)."""
    device = 0

    def pinned_empty(self, shape, dtype):
        return np.empty(shape, dtype)

    def pinned_free(self, a):
        pass

    def cnn_load(self, net_id, compiled):
        pass

    def set_signal(self, sig):
        assert sig.dtype == np.int16
        self.sig = np.array(sig, copy=True)

    def sidekit(self):
        with np.errstate(divide='ignore'):
            self.loge, self.mspec = osk.mfcc_mspec((self.sig / 32768.0).astype(np.float32))
        self.mspec = self.mspec.astype(np.float32)
        return len(self.loge)

    def get_loge(self):
        return self.loge.copy()

    def get_mspec(self):
        return self.mspec.copy()


def _write_wav(path, pcm):
    import struct
    pcm = np.asarray(pcm, dtype='<i2')
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + pcm.nbytes) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 16000, 32000, 2, 16) +
                b'data' + struct.pack('<I', pcm.nbytes))
        f.write(pcm.tobytes())


def test_super_batch_pipeline_equals_per_file_calls_with_fake_device(tmp_path):
    """pipeline.process_files on a fake device: files of uneven lengths (none a multiple of 160 samples) laid end to end in one
    buffer, ONE feature pass, ONE network call per net for all files, per-file Viterbi -- must give exactly what the per-file
    path gives on the same fake device (offset / frame / window-row / span bookkeeping of pipeline._Worker.run), a medium too
    short for the super-batch takes the single-file path, an undecodable one is a per-file error, results keep their index."""
    from inaspeechsegmenter_amd import pipeline

    def content_predict(nclass, salt):                               # depends on the window's values only (fake_predict.py
        def predict(batch):                                          # also mixes in the row index: no good across batchings)
            x = np.asarray(batch)
            q = np.floor(np.where(np.isfinite(x), x, 0).astype(np.float64) * 64.0).astype(np.int64)
            pref = ((q[:, 20:44, :, 0].sum(axis=(1, 2)) + salt * 1000) // 700) % nclass
            out = np.full((len(x), nclass), 0.002 / (nclass - 1))
            out[np.arange(len(x)), pref] = 0.998
            return out.astype(np.float32)
        return predict

    fake = _FakeDevice({0: content_predict(3, 1), 1: content_predict(2, 2)}, {0: 21, 1: 24})
    seg = object.__new__(S.Segmenter)
    seg.energy_ratio, seg.detect_gender, seg.ctx, seg.ffmpeg = 0.03, True, fake, None
    seg.vad, seg.gender = object.__new__(S.SpeechMusicNoise), object.__new__(S.Gender)
    seg.vad.ctx = seg.gender.ctx = fake
    seg.vad.compiled = seg.gender.compiled = None
    rng = np.random.default_rng(11)

    def medium(nsamp, seed):
        r = np.random.default_rng(seed)
        t = np.arange(nsamp) / 16000.0
        x = np.zeros(nsamp)
        pos = 0
        while pos < nsamp:                                           # silence / noise / harmonic / chord stretches
            n = int(r.uniform(0.4, 1.5) * 16000)
            kind = r.integers(0, 4)
            tt = t[pos:pos + n]
            if kind == 1:
                x[pos:pos + n] = r.normal(0, 0.03, len(tt))
            elif kind == 2:
                x[pos:pos + n] = 0.1 * sum(np.sin(2 * np.pi * 140 * k * tt) / k for k in range(1, 12)) * (1 + 0.5 * np.sin(2 * np.pi * 4 * tt))
            elif kind == 3:
                x[pos:pos + n] = 0.08 * sum(np.sin(2 * np.pi * f * tt) for f in (262, 330, 392))
            pos += n
        return np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)

    lengths = [70001, 123457, 40333, 9000, 98765, 160 * 300 + 1]   # 9000 samples: < 68 frames -> single-file path
    paths = []
    for i, n in enumerate(lengths):
        p = tmp_path / f'm{i}.wav'
        _write_wav(p, medium(n, 100 + i))
        paths.append(str(p))
    bad = tmp_path / 'broken.wav'
    bad.write_bytes(b'not a wav file')
    paths.insert(2, str(bad))

    got = {}

    def on_result(i, src, lseg, err, secs=0.0):
        assert i not in got and src == paths[i] and secs >= 0.0
        got[i] = (lseg, err)

    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pipeline.process_files(seg, paths, on_result, batch_files=3, workers=1, decode_threads=2)
        assert sorted(got) == list(range(len(paths)))
        assert got[2][0] is None and got[2][1].startswith('error')
        for i, p in enumerate(paths):
            if i == 2:
                continue
            from inaspeechsegmenter_amd.io import decode_pcm
            mspec, loge, difflen = S._sig2feats(fake, decode_pcm(p, None, None, None), p)
            want = seg.segment_feats(mspec, loge, difflen, 0)
            assert got[i][1] is None
            assert got[i][0] == want, (i, got[i][0][:4], want[:4])
    assert len({lab for lseg, _ in got.values() if lseg for lab, _, _ in lseg}) >= 3      # the inputs exercise several labels
    # Segmenter.dense_batches: both networks on every slot of every file, the rows of the segments picked afterwards -- the
    # same segments (bench.py's dense file-path figure is the same work the reference semantics could at most require)
    first = dict(got)
    got.clear()
    calls = []
    orig = fake.cnn_probs
    fake.cnn_probs = lambda net, rows: (calls.append(len(rows)), orig(net, rows))[1]
    seg.dense_batches = True
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pipeline.process_files(seg, paths, on_result, batch_files=3, workers=1, decode_threads=2)
    seg.dense_batches = False
    assert {i: v[0] for i, v in got.items() if i != 2} == {i: v[0] for i, v in first.items() if i != 2}
    nslots = [((n - 400) // 160 + 2) // 2 for n in lengths if n != 9000]                     # P = ceil(T / 2) slots per file
    assert sum(calls) >= 2 * sum(nslots)                                                      # every slot, both nets (+ the single-file medium)


def test_batch_process_contract_with_fake_device(tmp_path):
    """segmenter.py:297-335 through the product's batch_process + pipeline on a fake device: return tuple, message codes
    (0 ok / 1 already exists / 2 error) in INPUT order, output directories created, skipifexist, both exporters, unknown format
    (run_test.py:107-134 are the reference's checks of the same contract)."""
    import filecmp

    def predict3(batch):
        x = np.asarray(batch)
        out = np.full((len(x), 3), 0.001, np.float32)
        out[np.arange(len(x)), (np.nan_to_num(x[:, 30, 5, 0]) > 0).astype(int)] = 0.998
        return out

    def predict2(batch):
        x = np.asarray(batch)
        out = np.full((len(x), 2), 0.002, np.float32)
        out[np.arange(len(x)), (np.nan_to_num(x[:, 10, 3, 0]) > 0).astype(int)] = 0.998
        return out

    fake = _FakeDevice({0: predict3, 1: predict2}, {0: 21, 1: 24})
    seg = object.__new__(S.Segmenter)
    seg.energy_ratio, seg.detect_gender, seg.ctx, seg.ffmpeg = 0.03, True, fake, None
    seg.vad, seg.gender = object.__new__(S.SpeechMusicNoise), object.__new__(S.Gender)
    seg.vad.ctx = seg.gender.ctx = fake
    seg.vad.compiled = seg.gender.compiled = None
    src = os.path.join(GOLDEN, 'musanmix.wav')
    lout = [str(tmp_path / 'a' / 'b' / '1.csv'), str(tmp_path / '2.csv'), str(tmp_path / '3.csv'), str(tmp_path / '4.TextGrid')]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t, nb, avg, lmsg = seg.batch_process([src, os.path.join(GOLDEN, 'doesnotexist.wav'), src], lout[:3], workers=1)
        assert nb == 2 and t > 0 and abs(avg - t / 2) < 1e-9
        assert [m[0] for m in lmsg] == lout[:3] and [m[1] for m in lmsg] == [0, 2, 0]
        assert lmsg[1][2].startswith('error: ') and lmsg[0][2].startswith('ok ')
        assert filecmp.cmp(lout[0], lout[2], shallow=False) and not os.path.exists(lout[1])
        ref = tmp_path / 'ref.csv'
        export_funcs.seg2csv(seg.segment_signal(iss_io.decode_pcm(src, None, None, None)), str(ref))
        assert filecmp.cmp(lout[0], str(ref), shallow=False)
        t, nb, avg, lmsg = seg.batch_process([src, src], [lout[0], lout[2]], skipifexist=True, workers=1)
        assert nb == 0 and avg == -1 and [m[1:] for m in lmsg] == [(1, 'already exists')] * 2
        t, nb, avg, lmsg = seg.batch_process([src], [lout[3]], output_format='textgrid', workers=1)
        assert nb == 1 and open(lout[3]).read().startswith('File type = "ooTextFile"')
        with pytest.raises(NotImplementedError):
            seg.batch_process([src], [lout[0]], output_format='json', workers=1)
    seg.close()


def test_pipeline_device_failure_propagates(tmp_path):
    """A device failure (NativeError) inside a worker is not a per-file error: process_files re-raises it in the caller's
    thread (the reference lets predict() exceptions propagate, segmenter.py:314-327), and batch_process does not swallow it."""
    from inaspeechsegmenter_amd import pipeline

    class Broken(_FakeDevice):
        def cnn_probs(self, net_id, win_rows):
            raise _native.NativeError('iss_cnn_probs: out of memory (simulated)')

    fake = Broken({}, {0: 21, 1: 24})
    seg = object.__new__(S.Segmenter)
    seg.energy_ratio, seg.detect_gender, seg.ctx, seg.ffmpeg = 0.03, True, fake, None
    seg.vad, seg.gender = object.__new__(S.SpeechMusicNoise), object.__new__(S.Gender)
    seg.vad.ctx = seg.gender.ctx = fake
    seg.vad.compiled = seg.gender.compiled = None
    src = os.path.join(GOLDEN, 'musanmix.wav')
    seen = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        with pytest.raises(_native.NativeError, match='simulated'):
            pipeline.process_files(seg, [src, src], lambda *a: seen.append(a), workers=1)
        assert seen == []
        with pytest.raises(_native.NativeError, match='simulated'):
            seg.batch_process([src], [str(tmp_path / 'o.csv')], workers=1)
    assert not os.path.exists(tmp_path / 'o.csv')


def test_pipeline_failure_drains_every_stage(tmp_path):
    """With more super-batches queued than the bounded batch queue holds, a failing device worker must keep draining it:
    process_files raises instead of hanging with the packer blocked on `put` (one worker, and two workers failing both),
    and an exception raised by on_result (unwritable output) propagates the same way."""
    import threading
    from inaspeechsegmenter_amd import pipeline

    class Broken(_FakeDevice):
        def cnn_probs(self, net_id, win_rows):
            raise _native.NativeError('iss_cnn_probs: device lost (simulated)')

    def make(dev):
        seg = object.__new__(S.Segmenter)
        seg.energy_ratio, seg.detect_gender, seg.ctx, seg.ffmpeg = 0.03, True, dev, None
        seg.vad, seg.gender = object.__new__(S.SpeechMusicNoise), object.__new__(S.Gender)
        seg.vad.ctx = seg.gender.ctx = dev
        seg.vad.compiled = seg.gender.compiled = None
        return seg

    src = os.path.join(GOLDEN, 'musanmix.wav')
    out = {}

    def run(seg, cb, workers, batch_files=1):
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                pipeline.process_files(seg, [src] * 9, cb, batch_files=batch_files, workers=workers, decode_threads=2)
            out['exc'] = None
        except BaseException as e:                                  # noqa: B902
            out['exc'] = e

    for workers in (1, 2):
        seg = make(Broken({}, {0: 21, 1: 24}))
        if workers == 2:                                            # a second fake context instead of a real device one
            seg.__dict__['_pipeline_workers'] = [pipeline._Worker(seg, seg.ctx), pipeline._Worker(seg, seg.ctx)]
        th = threading.Thread(target=run, args=(seg, lambda *a: None, workers), daemon=True)
        th.start()
        th.join(60)
        assert not th.is_alive(), f'process_files hangs after a device failure ({workers} worker(s))'
        assert isinstance(out['exc'], _native.NativeError)

    def predict3(x):
        return np.tile(np.array([[.8, .1, .1]], np.float32), (len(x), 1))

    def predict2(x):
        return np.tile(np.array([[.3, .7]], np.float32), (len(x), 1))

    def cb(i, src, lseg, err, secs=0.0):
        raise OSError('disk full (simulated)')
    th = threading.Thread(target=run, args=(make(_FakeDevice({0: predict3, 1: predict2}, {0: 21, 1: 24})), cb, 1), daemon=True)
    th.start()
    th.join(120)
    assert not th.is_alive() and isinstance(out['exc'], OSError)

    # an exception inside the packer thread itself (not on_result, not a device call): it must be re-raised by process_files
    # instead of leaving files silently missing, with the decode threads drained
    class BadBatch(pipeline._Batch):
        def samples(self):
            raise MemoryError('packing failed (simulated)')
    orig = pipeline._Batch
    pipeline._Batch = BadBatch
    try:
        th = threading.Thread(target=run, args=(make(_FakeDevice({0: predict3, 1: predict2}, {0: 21, 1: 24})), lambda *a: None, 1, 4), daemon=True)
        th.start()
        th.join(120)
        assert not th.is_alive() and isinstance(out['exc'], MemoryError), out
    finally:
        pipeline._Batch = orig


def test_fused_energy_detector_equals_generic_path():
    """iss_energy_viterbi (comparison + pred2logemission + two-state Viterbi in one compiled call) == the generic
    `viterbi_decoding(pred2logemission(loge > threshold), log_trans_exp(150, cost0=-5))` of segmenter.py:69-73 -- random
    log-energies, -inf frames, all-silent input (NaN threshold), thresholds that equal samples."""
    rng = np.random.default_rng(0)
    for trial in range(120):
        T = int(rng.integers(1, 3000))
        loge = rng.normal(-5, 3, T).astype(np.float32)
        if trial % 5 == 0:
            loge[rng.integers(0, T, T // 3)] = -np.inf
        if trial % 17 == 0:
            loge[:] = -np.inf
        if trial % 7 == 0:
            loge = np.round(loge)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            thr = np.mean(loge[np.isfinite(loge)]) + np.log(0.03)
        assert isinstance(thr, np.float64)                      # float32 mean + float64 log: what the fast path keys on
        want = S.viterbi_decoding(S.pred2logemission(loge > thr), S.log_trans_exp(150, cost0=-5))
        got = S._energy_activity(loge, 0.03)
        assert got.dtype == want.dtype and np.array_equal(got, want), trial
        thr2 = np.float64(loge[0]) if np.isfinite(loge[0]) else thr
        assert np.array_equal(_native.energy_viterbi(loge, thr2, S._ENERGY_TRANS),
                              S.viterbi_decoding(S.pred2logemission(loge > thr2), S.log_trans_exp(150, cost0=-5))), trial


def test_oracle_is_only_used_as_the_checker():
    """oracle/ is test infrastructure: nothing in the product package, the scripts or the tools may import it; bench.py only
    inside its cpu_baseline / parity legs and __graft_entry__ only inside smoke()."""
    import ast

    def oracle_imports(path):
        tree = ast.parse(open(path).read())

        def is_oracle(n):
            if not isinstance(n, (ast.Import, ast.ImportFrom)):
                return False
            names = [a.name for a in n.names] if isinstance(n, ast.Import) else [n.module or '']
            return any(x == 'oracle' or x.startswith('oracle.') for x in names)

        inside = {}
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)):
                for n in ast.walk(fn):
                    if is_oracle(n):
                        inside[n.lineno] = fn.name            # (the innermost def wins: ast.walk reaches it last)
        hits = [(inside.get(n.lineno, '<module>'), n.lineno) for n in ast.walk(tree) if is_oracle(n)]
        return hits

    for d in ('inaspeechsegmenter_amd', 'scripts', 'tools'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith('.py'):
                    assert oracle_imports(os.path.join(dirpath, f)) == [], (dirpath, f)
    assert {fn for fn, _ in oracle_imports(os.path.join(ROOT, 'bench.py'))} <= {'cpu_baseline', '_cpu_feature_worker', '_cpu_full_worker', 'bench_vbx', 'parity_check'}
    assert {fn for fn, _ in oracle_imports(os.path.join(ROOT, '__graft_entry__.py'))} <= {'smoke'}



def test_asm_load_kernels_keep_their_ring_registers(tmp_path):
    """conv_pw.h issues its global loads from inline asm into tied register variables and waits for them with counted
    s_waitcnt statements the compiler knows nothing about: a compiler-inserted COPY of such a register between its load
    and the wait would read stale data.  Checked statically on the gfx950 disassembly of the unit (tools/check_ring_regs.py):
    the load destinations are only ever read by the conversion / epilogue arithmetic and LDS stores, only written by the
    loads and the initialisation, and nothing spills."""
    import shutil
    import subprocess
    if not (os.path.exists('/opt/rocm/bin/hipcc') and shutil.which('c++filt')):
        pytest.skip('no hipcc here')
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    r = subprocess.run(['bash', os.path.join(root, 'tools', 'kernel_meta.sh'), 'cnn_pw.hip', 'pws'], capture_output=True, text=True)
    assert r.returncode == 0 and 'spill v0 s0' in r.stdout, r.stdout + r.stderr
    assert all('spill v0 s0' in l for l in r.stdout.strip().splitlines()), r.stdout
    c = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_ring_regs.py'), '/tmp/iss_meta/cnn_pw.s', 'pws'],
                       capture_output=True, text=True)
    assert c.returncode == 0, c.stdout
    assert c.stdout.count("ok ") >= 7, c.stdout
    # the chained kernel (conv_pwc.h): its r-row registers are ring registers only across a tile boundary, so the per-load rule
    # applies -- the first instruction touching every load's destination is a consumer behind a wait (no AGPR parking)
    r = subprocess.run(['bash', os.path.join(root, 'tools', 'kernel_meta.sh'), 'cnn_pwc.hip', 'pwc'], capture_output=True, text=True)
    assert r.returncode == 0 and all('spill v0 s0' in l for l in r.stdout.strip().splitlines()), r.stdout + r.stderr
    c = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_ring_regs.py'), '/tmp/iss_meta/cnn_pwc.s', 'pwc'],
                       capture_output=True, text=True)
    assert c.returncode == 0 and c.stdout.count("ok ") == 5 and "BEFORE A WAIT" not in c.stdout, c.stdout


def test_bench_cpu_file_parallel_leg_runs_without_a_gpu():
    """bench.py's file-parallel CPU leg (leg 1b) spawns `bench.py --cpu-worker` processes that use numpy and
    the oracle only: it has to work on a host without a GPU or torch device, and a worker that dies is counted, not waited for."""
    import bench
    leg = bench.cpu_file_parallel_leg(2, nsec=4, budget_s=120.0)
    assert leg['processes'] == 2 and leg['finished'] == 2 and leg['failed'] == 0, leg
    assert leg['x_realtime_aggregate'] > 0 and leg['x_realtime_one_process_mean'] > 1.0, leg
    full = bench.cpu_full_path_all_cores_leg(4, threads=2, nsec=4, budget_s=240.0)               # leg 1c: 2 processes x 2 threads, CNNs included
    assert full['processes'] == 2 and full['finished'] == 2 and full['cores'] == 4 and full['x_realtime_aggregate_steady'] > 0, full
    pcm = bench.synth_recording_numpy(3, 5 * bench.FS)
    assert pcm.dtype == np.int16 and pcm.shape == (5 * bench.FS,) and np.abs(pcm).max() > 0
    assert np.array_equal(pcm, bench.synth_recording_numpy(3, 5 * bench.FS))          # seeded: the same file every time


def test_resnet_compile_marks_projection_blocks_for_the_two_source_gemm():
    """compile_resnet101 leaves the projection shortcut and the expansion as two rows and adds, on the second, the blob offsets
    (+ 1) of the concatenated matrix [W_exp | W_proj] and the summed bias (include/iss.h ISS_C_DUALW / ISS_C_DUALB): one per stage."""
    from inaspeechsegmenter_amd import keras_model as KM, _native as N
    comp = KM.compile_resnet101(KM.synthetic_resnet101(3), 64, 144, window_input=True)
    prog = np.asarray(comp.prog).reshape(-1, N.PROG_COLS)
    blob = np.asarray(comp.blob)
    rows = [i for i, r in enumerate(prog) if r[N.C_DUALW] > 0]
    assert len(rows) == 4
    for i in rows:
        r, p = prog[i], prog[i - 1]
        assert r[N.C_RES] == p[N.C_OUT] == r[N.C_OUT] and p[N.C_ACT] == 0 and r[N.C_ACT] == 1 and p[N.C_KH] == r[N.C_KH] == 1
        k = r[N.C_CIN] + p[N.C_CIN]
        W = blob[r[N.C_DUALW] - 1:r[N.C_DUALW] - 1 + r[N.C_COUT] * k].reshape(r[N.C_COUT], k)
        We = blob[r[N.C_WOFF]:r[N.C_WOFF] + r[N.C_COUT] * r[N.C_CIN]].reshape(r[N.C_COUT], -1)
        Wp = blob[p[N.C_WOFF]:p[N.C_WOFF] + p[N.C_COUT] * p[N.C_CIN]].reshape(p[N.C_COUT], -1)
        assert np.array_equal(W, np.concatenate([We, Wp], axis=1)) and (r[N.C_DUALW] - 1) % 8 == 0
        b = blob[r[N.C_DUALB] - 1:r[N.C_DUALB] - 1 + r[N.C_COUT]]
        assert np.array_equal(b, blob[r[N.C_BOFF]:r[N.C_BOFF] + r[N.C_COUT]] + blob[p[N.C_BOFF]:p[N.C_BOFF] + r[N.C_COUT]])
    assert all(r[N.C_DUALW] == 0 and r[N.C_DUALB] == 0 for i, r in enumerate(prog) if i not in rows)


def test_every_diag_switch_of_the_abi_has_its_python_name():
    """include/iss.h ISS_DIAG_* <-> _native.DIAG_BITS: the same bits under the lower-case names without the prefix (what the
    ISS_DIAG environment variable and Context.set_diag take), ISS_DIAG_ALL = their union, and unknown names are refused."""
    import re
    from inaspeechsegmenter_amd import _native
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
    defs = dict(re.findall(r'#define\s+ISS_DIAG_(\w+)\s+(0x[0-9a-fA-F]+)u', open(os.path.join(root, 'include', 'iss.h')).read()))
    allbits = int(defs.pop('ALL'), 16)
    assert {k.lower(): int(v, 16) for k, v in defs.items()} == _native.DIAG_BITS
    union = 0
    for v in _native.DIAG_BITS.values():
        assert v and v & (v - 1) == 0 and not (union & v)               # one bit each, no bit twice
        union |= v
    assert union == allbits
    assert _native.diag_flags('no_ring,no_fsame+no_gfused') == 0x2000 | 0x4000 | 0x40000
    with pytest.raises(KeyError):
        _native.diag_flags('no_such_switch')


def test_one_arithmetic_decision_per_network_across_device_contexts():
    """The precision guard (include/iss.h) decides a network's arithmetic at its first call PER CONTEXT; `DnnSegmenter.probs`
    makes the first call anywhere decide for every device context of the Segmenter (its own and the pipeline workers'): later
    contexts are told the outcome before their first call and are never probed; with the guard off nothing is touched."""
    from inaspeechsegmenter_amd import segmenter as S

    class Ctx:
        def __init__(self, outcome):
            self.outcome, self.state, self.mode, self.calls, self.told = outcome, 'pending', 'f16x3', 0, []

        def cnn_probs(self, net_id, rows):
            self.calls += 1
            if self.state == 'pending' and self.outcome is not None:
                self.state, self.mode = self.outcome
            return np.zeros((len(rows), 2), np.float32), np.ones(len(rows), np.uint8)

        def cnn_precision_info(self, net_id):
            return {'state': self.state, 'mode': self.mode, 'max_dlogp': None, 'max_dlogp_in_use': None, 'slots': 0}

        def cnn_set_net_precision(self, net_id, mode):
            self.told.append(mode); self.state = 'fixed'
            self.mode = {_native.PREC_BF16X3: 'bf16x3', _native.PREC_F32: 'f32', _native.PREC_F16X3: 'f16x3'}[mode]

    net = S.DnnSegmenter.__new__(S.Gender)
    a, b, c = Ctx(('escalated', 'bf16x3')), Ctx(('passed', 'f16x3')), Ctx(('escalated', 'f32'))
    rows = np.arange(4, dtype=np.int32)
    net.probs(a, rows)                                   # the first call anywhere: the library probes and decides
    net.probs(b, rows); net.probs(c, rows); net.probs(b, rows); net.probs(a, rows)
    assert a.told == [] and b.told == [_native.PREC_BF16X3] and c.told == [_native.PREC_BF16X3]
    assert (a.mode, b.mode, c.mode) == ('bf16x3', 'bf16x3', 'bf16x3') and (a.calls, b.calls, c.calls) == (2, 2, 1)
    off = S.DnnSegmenter.__new__(S.Gender)               # guard off: the state stays 'pending', nobody is told anything
    d, e = Ctx(None), Ctx(None)
    off.probs(d, rows); off.probs(e, rows); off.probs(d, rows)
    assert d.told == [] and e.told == [] and d.calls == 2 and e.calls == 1
