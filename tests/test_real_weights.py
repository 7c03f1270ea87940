"""T3: whole-pipeline parity with the reference's golden CSV / TextGrid files.
Needs the real Keras model files (release assets, remote_utils.py:4-15) under
~/.keras/inaSpeechSegmenter/ -- as .hdf5 (with h5py importable) or as the .npz export of
tools/convert_keras_hdf5.py.  They cannot be downloaded in the build environment, so these tests skip
themselves there; they are the tests that pin the CNN forward against TensorFlow's results
(run_test.py:90-127 test_processingresult / test_batch / test_praat_export)."""
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
