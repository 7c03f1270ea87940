"""GPU parity, config 5: VBx 64-band fbank front end and the ResNet-101 x-vector network through
the C ABI, against the committed reference outputs (media/test.h5, reference get_features) and
the oracle.  Tolerances: features 2e-5 abs (float64 pipeline, float32 result: FFT rounding differs
from pocketfft at 1e-16, a few results land on the other side of a float32 rounding boundary);
embeddings 1e-4 of the embedding scale (measured 1e-5; the reference holds its ONNX x-vector to 4 decimals, run_test.py:189-195)."""
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import vbx as V, keras_model as KM
from oracle import vbx as ovbx
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
FEA_TOL = 2e-5


@pytest.fixture(scope='module')
def golden_vbx():
    return np.load(os.path.join(GOLDEN, 'vbx_feats.npz'))


def test_features_match_reference_outputs(ctx, golden_vbx):
    g = golden_vbx
    sig = g['lamartine_pcm16'].astype(np.float64) / 32768.0           # what ffmpeg's pcm_s16le hop yields
    fea = V.FeatureExtractor(ctx)(sig)
    assert fea.shape == g['lamartine_fea'].shape and fea.dtype == np.float32
    err = np.abs(fea - g['lamartine_fea']).max()
    same = np.mean(fea == g['lamartine_fea'])
    print(f'lamartine: max abs err {err:.2e}, {same * 100:.2f}% bit-identical')
    assert err <= FEA_TOL
    assert same > 0.9
    assert np.abs(fea[:144] - g['test_h5_melbands']).max() <= FEA_TOL      # run_test.py:189-195 input fixture


@pytest.mark.parametrize('n', [200, 1000, 16000, 48000 + 37, 160 * 301])
def test_features_ragged_lengths_vs_oracle(ctx, n):
    """n < 300 frames exercises the global-mean branch of cmvn_floating_kaldi (features_vbx.py:143)."""
    rng = np.random.default_rng(n)
    sig = np.clip(rng.normal(0, 0.1, n), -1, 1)
    fea = V.FeatureExtractor(ctx)(sig)
    ref = ovbx.get_features(sig)
    assert fea.shape == ref.shape
    assert np.abs(fea - ref).max() <= FEA_TOL


@pytest.fixture(scope='module')
def extractor(ctx):
    return V.VBxExtractor(ctx, ovbx.resnet101_random_params(0), batch_windows=8)


def test_resnet101_matches_reference_topology_golden(extractor):
    """tests/golden/resnet_golden.npz was produced by the reference's own resnet.py (torch-CPU) with the
    oracle's seeded parameters: pins the topology/lowering (Bottleneck [3,4,23,3], stats pooling)."""
    g = np.load(os.path.join(GOLDEN, 'resnet_golden.npz'))
    x = g['x']                                                        # (2, 64, 144) feature-major
    fea = [xi.T for xi in x]                                          # (144, 64) each, as VBxExtractor sees them
    emb = np.stack([extractor.get_embedding(f) for f in fea])
    scale = np.abs(g['emb']).max()
    assert np.abs(emb - g['emb']).max() <= 1e-4 * scale, (np.abs(emb - g['emb']).max(), scale)


def test_window_loop_and_tail_window(extractor):
    rng = np.random.default_rng(5)
    T = 144 + 24 * 3 + 17                                             # 3 full windows + a 65-frame tail
    fea = rng.normal(0, 1, (T, 64)).astype(np.float32)
    got = extractor('utt', fea, T / 100.0)
    wins = ovbx.window_list(T)
    assert [k for k, _, _ in got] == [f'utt_{a:08}-{b:08}' for a, b in wins]
    assert got[0][1] == (0.0, 1.44) and got[-1][1] == (round(wins[-1][0] / 100.0, 3), round(T / 100.0, 3))
    for (key, seg, x), (a, b) in zip(got, wins):
        ref = ovbx.resnet101_forward(extractor.params, fea[a:b].T[None])[0] * 10
        assert np.abs(x - ref).max() <= 1e-4 * np.abs(ref).max(), key


def test_batched_equals_single(extractor):
    rng = np.random.default_rng(6)
    fea = rng.normal(0, 1, (144 + 24 * 10, 64)).astype(np.float32)
    starts = list(range(0, 24 * 10, 24))
    a = extractor.get_embeddings(fea, starts, 144)
    b = np.stack([extractor.get_embedding(fea[s:s + 144]) for s in starts])
    assert np.array_equal(a, b)


def test_projection_shortcut_and_expansion_as_one_gemm(ctx, extractor):
    """The first Bottleneck of each stage (resnet.py:60-75: relu(bn3(conv3(r)) + bn_s(conv_s(x)))) runs as ONE two-source GEMM
    over [r | x at the block's stride] on the concatenated weights (ISS_C_DUALW, conv_x3_pws2_kernel<.., DUAL>): same
    x-vectors as the two launches it replaces (the f32 sum is re-associated: 1e-5 of the embedding scale), at 144 frames and
    on a tail window, four such launches per pass, none with the switch off; the exact-f32 mode never takes it."""
    rng = np.random.default_rng(9)
    for frames, nwin in ((144, 11), (65, 3)):
        fea = rng.normal(0, 1, (frames + 24 * nwin, 64)).astype(np.float32)
        starts = list(range(0, 24 * nwin, 24))
        ctx.prof_enable(True)
        ctx.prof_reset()
        a = extractor.get_embeddings(fea, starts, frames)
        inst = {e['kernel']: e['launches'] for e in ctx.prof_instances()}
        ctx.set_diag('no_dual')
        try:
            ctx.prof_reset()
            b = extractor.get_embeddings(fea, starts, frames)
            inst_off = {e['kernel']: e['launches'] for e in ctx.prof_instances()}
        finally:
            ctx.set_diag(0)
            ctx.prof_enable(False)
        dual = [k for k in inst if 'dual' in k]
        passes = -(-nwin // extractor.batch_windows)
        assert dual and inst[dual[0]] == 4 * passes, inst
        assert not [k for k in inst_off if 'dual' in k], inst_off
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 1e-5 * scale, (frames, np.abs(a - b).max(), scale)
        ref = np.stack([ovbx.resnet101_forward(extractor.params, fea[s:s + frames].T[None])[0] for s in starts[:3]])
        assert np.abs(a[:3] - ref).max() <= 1e-4 * np.abs(ref).max()
    from inaspeechsegmenter_amd import _native as N
    ctx.set_precision(N.PREC_F32)
    try:
        ctx.prof_enable(True)
        ctx.prof_reset()
        c = extractor.get_embeddings(fea, starts, frames)
        assert not [e for e in ctx.prof_instances() if 'dual' in e['kernel']]
        assert np.abs(c - a).max() <= 1e-4 * np.abs(a).max()
    finally:
        ctx.prof_enable(False)
        ctx.set_precision(N.PREC_BF16X3)


def test_expansion_and_next_reduction_as_one_chained_launch(ctx, extractor):
    """conv_x3_pwc_kernel: the in-place 1x1 expansion of an identity Bottleneck (+ residual, relu) and the next Bottleneck's 1x1
    reduction (to 32 / 64 / 128 channels) in one launch -- the reduction reads x' out of LDS instead of HBM.  Same x-vectors as the
    two launches (1e-5 of the embedding scale: the first GEMM sums its k-steps in two accumulators), on full and tail windows and
    on a window count that leaves a partial last row tile; 26 such launches per pass (stages 1-3 and the transitions between them)."""
    rng = np.random.default_rng(10)
    for frames, nwin in ((144, 5), (65, 3), (144, 1)):
        fea = rng.normal(0, 1, (frames + 24 * nwin, 64)).astype(np.float32)
        starts = list(range(0, 24 * nwin, 24))
        ctx.prof_enable(True)
        ctx.prof_reset()
        a = extractor.get_embeddings(fea, starts, frames)
        inst = {e['kernel']: e['launches'] for e in ctx.prof_instances()}
        ctx.set_diag('no_chain')
        try:
            ctx.prof_reset()
            b = extractor.get_embeddings(fea, starts, frames)
            inst_off = {e['kernel']: e['launches'] for e in ctx.prof_instances()}
        finally:
            ctx.set_diag(0)
            ctx.prof_enable(False)
        want = {'conv_x3_pwc_kernel<4,4>': 21, 'conv_x3_pwc_kernel<2,4>': 1, 'conv_x3_pwc_kernel<2,2>': 2,
                'conv_x3_pwc_kernel<1,2>': 1, 'conv_x3_pwc_kernel<1,1>': 1}       # <C1 / 32, C3 / 32>: stages 3, 2 -> 3, 2, 1 -> 2, 1
        assert {k: v for k, v in inst.items() if 'pwc' in k} == want, inst
        assert not [k for k in inst_off if 'pwc' in k], inst_off
        scale = np.abs(b).max()
        assert np.abs(a - b).max() <= 1e-5 * scale, (frames, nwin, np.abs(a - b).max(), scale)
        ref = np.stack([ovbx.resnet101_forward(extractor.params, fea[s:s + frames].T[None])[0] for s in starts[:2]])
        assert np.abs(a[:2] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_pcm16_path_and_device_window_gather(ctx, extractor, golden_vbx):
    """PCM16 entry + cached dither == the int32 + per-call dither entry (bit-identical); x-vectors from the
    device-side window gather (iss_vbx_embed) == the host-stacked windows through iss_cnn_forward."""
    pcm = golden_vbx['lamartine_pcm16'][:16000 * 6]
    fe = V.FeatureExtractor(ctx)
    a = fe(pcm)                                                        # int16 -> iss_vbx_features_pcm16
    b = ctx.vbx_features(pcm.astype(np.int32), V.dither_stream(len(pcm)))
    assert np.array_equal(a, b)
    assert np.array_equal(fe(pcm.astype(np.float64) / 32768.0), a)     # float input that is PCM16-exact takes the same path
    assert np.array_equal(fe(pcm[:16000 * 2]), ctx.vbx_features(pcm[:32000].astype(np.int32), V.dither_stream(32000)))   # cached prefix
    fea = fe(pcm)
    got = extractor('utt', fea, len(pcm) / 16000.0)                    # resident features -> device windows
    want = extractor('utt', fea.copy(), len(pcm) / 16000.0)            # a copy is not the resident array -> host path
    assert [g[0] for g in got] == [w[0] for w in want] and len(got) > 15
    for (k, seg, x), (_, seg2, y) in zip(got, want):
        assert seg == seg2 and np.abs(x - y).max() <= 1e-5 * max(1.0, np.abs(y).max()), k
    for n in (16000 * 3 + 777, 16000 * 4 + 99, 16000 * 5 + 1234):     # three more tail lengths: the two tail slots recycle
        f2 = fe(pcm[:n])
        assert len(extractor('u', f2, n / 16000.0)) == len(ovbx.window_list(len(f2)))


def test_resident_features_handle(ctx, extractor, golden_vbx):
    """FeatureExtractor(..., to_host=False): the (T, 64) array stays in HBM, the x-vectors are the same as with the copy."""
    fe = V.FeatureExtractor(ctx)
    pcm = golden_vbx['lamartine_pcm16']
    fea = fe(pcm)
    a = extractor('utt', fea, len(pcm) / 16000.0)
    h = fe(pcm, to_host=False)
    assert isinstance(h, V.ResidentFeatures) and len(h) == len(fea)
    b = extractor('utt', h, len(pcm) / 16000.0)
    assert [x[:2] for x in a] == [x[:2] for x in b] and all(np.array_equal(x[2], y[2]) for x, y in zip(a, b))


def test_final_onnx_file_through_the_product_loader(ctx, tmp_path, monkeypatch):
    """The reference's live backend loads `final.onnx` (vbx_segmenter.py:249-266).  A file of that shape written here (seeded
    ResNet-101, BatchNorm folded the way torch's exporter does) placed where `get_remote` would put it must be found by
    locate_model, read by the package's own protobuf walk (onnx_reader.py) and give the oracle's x-vectors on the device."""
    from test_onnx_reader import write_resnet_onnx
    from inaspeechsegmenter_amd import segmenter as S
    from inaspeechsegmenter_amd.vfs import _load_resnet_params
    params = KM.synthetic_resnet101(5)
    d = tmp_path / 'inaSpeechSegmenter'
    d.mkdir()
    write_resnet_onnx(str(d / 'final.onnx'), params, folded=True, tensor_style='raw')
    monkeypatch.setattr(S, '_MODEL_DIRS', [str(d)])
    got_params = _load_resnet_params(S.locate_model('final.onnx'))
    ex = V.VBxExtractor(ctx, got_params)
    rng = np.random.default_rng(5)
    fea = rng.normal(0, 1, (3 * 144, 64)).astype(np.float32)
    starts = [0, 100, 288]
    got = ex.get_embeddings(fea, starts, 144)
    want = ovbx.resnet101_forward(params, np.stack([fea[s:s + 144].T for s in starts]))
    assert np.abs(got - want).max() <= 1e-4 * max(1.0, np.abs(want).max())
