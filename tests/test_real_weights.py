"""T3: whole-pipeline parity with the reference's golden CSV / TextGrid files.
Needs the real Keras model files (release assets, remote_utils.py:4-15) under
~/.keras/inaSpeechSegmenter/ -- as .hdf5 (with h5py importable) or as the .npz export of
tools/convert_keras_hdf5.py.  They cannot be downloaded in the build environment, so these tests skip
themselves there; they are the tests that pin the CNN forward against TensorFlow's results
(run_test.py:90-127 test_processingresult / test_batch / test_praat_export).

Round 4 adds what closes row a11 of SURVEY.md section 8 the day weights exist:
  * frame logits against a dump of the reference's own `nn.predict` outputs (tools/dump_reference_outputs.py, run wherever
    TensorFlow + the model files are; fixture tests/golden/real_reference_dump.npz) -- log-probabilities within 1e-3, the
    north star's bound;
  * the x-vector of media/test.h5:lamartinemelbands against media/test.h5:lamartineonnx to 4 decimals (run_test.py:189-195
    test_vbx_onnx; both arrays are already committed in tests/golden/vbx_feats.npz) -- needs final.onnx only, read by
    inaspeechsegmenter_amd/onnx_reader.py;
  * VoiceFemininityScoring('vfp') on media/lamartine.wav == 0.534884 to 6 decimals (run_test.py:177-187 test_vf_score)."""
import filecmp
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _have(fname):
    from inaspeechsegmenter_amd.segmenter import locate_model
    try:
        locate_model(fname)
        return True
    except FileNotFoundError:
        return False


need_smn = pytest.mark.skipif(not (_have('keras_speech_music_noise_cnn.hdf5') and _have('keras_male_female_cnn.hdf5')),
                              reason='real smn/gender Keras weights not installed')
need_sm = pytest.mark.skipif(not (_have('keras_speech_music_cnn.hdf5') and _have('keras_male_female_cnn.hdf5')),
                             reason='real sm/gender Keras weights not installed')


@need_smn
def test_musanmix_smn_gender_csv_and_textgrid_byte_identical(tmp_path):
    from inaspeechsegmenter_amd import Segmenter
    seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None)
    src = os.path.join(GOLDEN, 'musanmix.wav')
    t, nb, avg, lmsg = seg.batch_process([src], [str(tmp_path / 'a.csv')])
    assert nb == 1
    assert filecmp.cmp(str(tmp_path / 'a.csv'), os.path.join(GOLDEN, 'musanmix-smn-gender.csv'), shallow=False)
    seg.batch_process([src], [str(tmp_path / 'a.TextGrid')], output_format='textgrid')
    assert filecmp.cmp(str(tmp_path / 'a.TextGrid'), os.path.join(GOLDEN, 'musanmix-smn-gender.TextGrid'), shallow=False)


@need_sm
def test_musanmix_sm_gender_matches_golden(tmp_path):
    from inaspeechsegmenter_amd import Segmenter
    seg = Segmenter(vad_engine='sm', detect_gender=True, ffmpeg=None)
    res = seg(os.path.join(GOLDEN, 'musanmix.wav'))
    rows = [l.rstrip('\n').split('\t') for l in open(os.path.join(GOLDEN, 'musanmix-sm-gender.csv'))][1:]
    assert [r[0] for r in rows] == [l for l, _, _ in res]
    np.testing.assert_almost_equal([float(r[1]) for r in rows], [s for _, s, _ in res])
    np.testing.assert_almost_equal([float(r[2]) for r in rows], [e for _, _, e in res])


DUMP = os.path.join(GOLDEN, 'real_reference_dump.npz')
need_dump = pytest.mark.skipif(not os.path.exists(DUMP), reason='tests/golden/real_reference_dump.npz missing: run tools/dump_reference_outputs.py '
                                                                 'where TensorFlow and the model files exist')
need_onnx = pytest.mark.skipif(not (_have('final.onnx') or _have('raw_81.pth')), reason='x-vector weights (final.onnx / raw_81.pth) not installed')
need_vfs = pytest.mark.skipif(not (_have('keras_speech_music_noise_cnn.hdf5') and (_have('final.onnx') or _have('raw_81.pth'))
                                   and _have('interspeech2023_cvfr.hdf5')), reason='VFS model files (smn VAD, final.onnx, interspeech2023_cvfr) not installed')


@need_smn
@need_dump
@pytest.mark.parametrize('engine', ['smn', 'gender'])
def test_frame_logits_match_the_reference_dump(engine):
    """iss_cnn_probs with the REAL weights on the slots the reference evaluated for media/musanmix.wav vs the reference's own
    `nn.predict` rows: same arg-max everywhere, |log p - log p_ref| <= 1e-3 (north star), |p - p_ref| <= 2e-4."""
    from inaspeechsegmenter_amd import Segmenter
    from inaspeechsegmenter_amd import segmenter as S
    from inaspeechsegmenter_amd.io import decode_pcm
    z = np.load(DUMP)
    seg = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None)
    pcm = decode_pcm(os.path.join(GOLDEN, 'musanmix.wav'), ffmpeg=None)
    seg.segment_signal(pcm)                                    # leaves the features of the file resident
    rows = S._window_rows(seg.ctx.T)
    slots = z[f'{engine}_musanmix_batch_slots']
    want = z[f'{engine}_musanmix_rawpred']
    net = seg.vad if engine == 'smn' else seg.gender
    got, fin = seg.ctx.cnn_probs(net.net_id, rows[slots])
    ok = fin & np.all(np.isfinite(want), axis=1)
    assert ok.sum() > 0.9 * len(slots)
    assert np.array_equal(got[ok].argmax(1), want[ok].argmax(1))
    with np.errstate(divide='ignore'):
        dlog = np.abs(np.log(got[ok]) - np.log(want[ok]))
    dlog = dlog[np.isfinite(dlog)]
    print(f'{engine}: {ok.sum()} slots, max |dp| {np.abs(got[ok] - want[ok]).max():.2e}, max |dlogp| {dlog.max():.2e}')
    assert np.abs(got[ok] - want[ok]).max() <= 2e-4
    assert dlog.max() <= 1e-3


@need_smn
@need_dump
def test_topology_of_the_real_nets_is_lowered_without_fallbacks():
    """The parsed model_config of the dump == the installed file's, and every conv / dense row of the lowered program is
    one the GEMM kernels take (no unknown op)."""
    import json
    from inaspeechsegmenter_amd import keras_model as KM
    from inaspeechsegmenter_amd.segmenter import locate_model
    z = np.load(DUMP)
    for engine, fname in (('smn', 'keras_speech_music_noise_cnn.hdf5'), ('gender', 'keras_male_female_cnn.hdf5')):
        layers, shp = KM.load_model_file(locate_model(fname))
        cfg = json.loads(str(z[f'{engine}_model_config']))
        names = [l['class_name'] for l in cfg['config']['layers']]
        assert len([n for n in names if n in ('Conv2D', 'Dense')]) == len([l for l in layers if l['type'] in ('conv2d', 'dense')])
        KM.compile_layers(layers, shp)


@need_onnx
def test_vbx_onnx_xvector_of_the_reference_fixture():
    """run_test.py:189-195 test_vbx_onnx: the (144, 64) mel bands of media/test.h5 through the real ResNet-101 == the stored
    x-vector to 4 decimals (both arrays are committed in tests/golden/vbx_feats.npz)."""
    from inaspeechsegmenter_amd import _native
    from inaspeechsegmenter_amd.segmenter import locate_model
    from inaspeechsegmenter_amd.vbx import VBxExtractor
    from inaspeechsegmenter_amd.vfs import _load_resnet_params
    z = np.load(os.path.join(GOLDEN, 'vbx_feats.npz'))
    feats, ref = z['test_h5_melbands'], z['test_h5_onnx']
    try:
        path = locate_model('final.onnx')
    except FileNotFoundError:
        path = locate_model('raw_81.pth')
    ctx = _native.Context(0)
    ex = VBxExtractor(ctx, _load_resnet_params(path))
    got = ex.get_embedding(feats)
    np.testing.assert_almost_equal(ref, got, decimal=4)


@need_vfs
def test_vf_score_of_lamartine():
    """run_test.py:177-187 test_vf_score."""
    from inaspeechsegmenter_amd.vfs import VoiceFemininityScoring
    vfs = VoiceFemininityScoring(gd_model_criteria='vfp', ffmpeg=None)
    np.testing.assert_almost_equal(vfs(os.path.join(GOLDEN, 'lamartine.wav'))[0], 0.534884, decimal=6)
