"""CPU: the oracle restatement against the committed reference outputs (tests/golden, produced by
tests/golden/make_golden.py running the real reference modules) and the reference's own golden
CSV files."""
import os

import numpy as np

from oracle import sidekit as osk, viterbi as ovit, segment as oseg, vbx as ovbx
from conftest import GOLDEN, read_wav_int16, synth_pcm


def _sig(name):
    pcm = read_wav_int16(os.path.join(GOLDEN, name))
    return (pcm / 32768.0).astype(np.float32)


def test_melbank_matches_reference():
    bank, _ = osk.mel_bank()
    ref = np.load(os.path.join(GOLDEN, 'sidekit_melbank.npy'))
    assert np.array_equal(bank, ref)
    assert (bank != 0).sum() == 454


def test_sidekit_bit_exact_vs_reference():
    g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    cases = {'musanmix': _sig('musanmix.wav'), 'silence': _sig('silence2sec.wav'),
             'synth': (synth_pcm(1234, 48000) / 32768.0).astype(np.float32),
             'short': (synth_pcm(77, 16000)[7000:17000] / 32768.0).astype(np.float32)}
    for tag, sig in cases.items():
        loge, mspec = osk.mfcc_mspec(sig)
        assert np.array_equal(loge, g[tag + '_loge'], equal_nan=True), tag
        assert np.array_equal(mspec, g[tag + '_mspec'], equal_nan=True), tag


def test_viterbi_known_answers():
    g = np.load(os.path.join(GOLDEN, 'viterbi_cases.npz'))
    for i in range(6):
        assert np.array_equal(ovit.viterbi_decoding(g[f'em{i}'], g[f'tr{i}']), g[f'st{i}'])


def _csv_rows(path):
    rows = [l.rstrip('\n').split('\t') for l in open(path)][1:]
    return [(r[0], float(r[1]), float(r[2])) for r in rows]


def test_energy_boundaries_reproduce_reference_csv():
    """All noEnergy rows and every row boundary of media/musanmix-smn-gender.csv are weight-free."""
    g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    lseg = oseg.energy_seglist(g['musanmix_loge'], 0.03)
    gold = _csv_rows(os.path.join(GOLDEN, 'musanmix-smn-gender.csv'))
    mine_ne = [(s * .02, e * .02) for lab, s, e in lseg if lab == 'noEnergy']
    gold_ne = [(s, e) for lab, s, e in gold if lab == 'noEnergy']
    assert mine_ne == gold_ne
    bounds = sorted({0 + s * .02 for _, s, _ in lseg} | {0 + e * .02 for _, _, e in lseg})
    gold_bounds = sorted({s for _, s, _ in gold} | {e for _, _, e in gold})
    assert bounds == gold_bounds


def test_silence_csv_weight_free():
    g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    never = lambda b: (_ for _ in ()).throw(AssertionError('no slot may reach the CNN'))
    res = oseg.segment_feats(g['silence_mspec'], g['silence_loge'], 0, 0, 'smn', never, never)
    assert res == _csv_rows(os.path.join(GOLDEN, 'silence2sec-smn-gender.csv'))


def test_short_media_slot_count():
    """media/0021-smn-gender.csv: 66 frames -> 33 slots -> stop 0.66 (weights only pick the label)."""
    m = np.random.default_rng(0).normal(0, 1, (66, 24)).astype(np.float32)
    mspec = np.concatenate((m, np.ones((2, 24)) * np.min(m)))
    pred = lambda b: np.tile(np.array([[0.1, 0.9]], np.float32), (len(b), 1))
    lseg = oseg.dnn_segment('gender', pred, mspec, [('speech', 0, 33)], difflen=2)
    assert lseg == [('male', 0, 33)] and 33 * .02 == 0.66


def test_vbx_features_match_reference_h5():
    g = np.load(os.path.join(GOLDEN, 'vbx_feats.npz'))
    sig = g['lamartine_pcm16'].astype(np.float64) / 32768
    fea = ovbx.get_features(sig)
    assert np.array_equal(fea, g['lamartine_fea'])
    assert np.array_equal(fea[:144], g['test_h5_melbands'])      # media/test.h5, run_test.py:189-195
    assert ovbx.window_list(len(fea))[-1][1] == len(fea)


def test_segmentation_bookkeeping_pinned_against_reference_lines(golden):
    """oracle/segment.py (get_patches, dnn_segment, segment_feats, binidx2seglist) == what the reference's own lines
    (segmenter.py:53-108,135-179,250-276, ast-extracted and executed by tests/golden/ref_segmenter_pin.py under the
    skimage interpreter) produced: bit-identical patches / finite masks, identical labels and boundaries."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(golden, 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.check_segmenter_pin(np.load(os.path.join(golden, 'segmenter_pin.npz')), np.load(os.path.join(golden, 'sidekit_feats.npz')))
