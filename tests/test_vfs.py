"""Voice-femininity tail.  CPU: the interval logic against a 1 ms raster; GPU: the whole
VoiceFemininityScoring pipeline with seeded stand-in weights against the oracle's features / ResNet."""
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import vfs
from conftest import GOLDEN


def _raster(speech, n=20000):
    r = np.zeros(n, bool)
    for s, e in speech:
        r[int(round(s * 1000)):int(round(e * 1000))] = True
    return r


def test_interval_logic_against_raster():
    rng = np.random.default_rng(8)
    for _ in range(50):
        cuts = np.sort(rng.choice(np.arange(1, 19999), 8, replace=False)) / 1000.0
        speech = [(cuts[0], cuts[1]), (cuts[2], cuts[3]), (cuts[4], cuts[5]), (cuts[6], cuts[7])]
        r = _raster(speech)
        for _ in range(20):
            a = rng.integers(0, 18000) / 1000.0
            b = a + rng.integers(2, 1500) * 2 / 1000.0            # even length in ms -> midpoint on the 1 ms grid
            ov = r[int(round(a * 1000)):int(round(b * 1000))].sum() / 1000.0
            assert abs(vfs.overlap_ratio(a, b, speech) - ov / (b - a)) < 1e-9
            m = int(round((a + b) / 2 * 1000))
            inside = bool(r[m]) and bool(r[m - 1])                  # strictly inside: not on a boundary
            boundary = any(abs((a + b) / 2 - x) < 1e-9 for seg in speech for x in seg)
            if not boundary:
                assert vfs.is_mid_speech(a, b, speech) == inside


def test_apply_vad_and_fallback_rule():
    speech = [(0.0, 10.0)]
    x = np.zeros(4)
    xs = [(f'k{i}', (i * 0.24, i * 0.24 + 1.44), x) for i in range(30)]
    kept = vfs.apply_vad(xs, speech, 0.7)
    assert [k for k, _, _ in kept] == [f'k{i}' for i in range(30)]            # every window lies inside speech
    kept = vfs.apply_vad(xs, [(0.0, 3.0)], 0.7)
    assert [k for k, _, _ in kept] == [f'k{i}' for i in range(9)]             # k8 = (1.92, 3.36): 75 % overlap; k9 = 58 %: dropped
    # windows straddling the end of speech: midpoint inside but overlap < threshold -> dropped, unless fewer than
    # round(50 %) of the mid-in-speech windows survive (vbx_segmenter.py:40-52)
    speech2 = [(0.0, 0.9)]
    xs2 = [('a', (0.0, 1.44), x), ('b', (0.1, 1.54), x), ('c', (0.2, 1.64), x)]
    kept2 = vfs.apply_vad(xs2, speech2, 0.7)     # a, b have their midpoint in speech (ratios .625, .556 < .7), c has not:
    assert [k for k, _, _ in kept2] == ['a']       # min_pred = round(0.5 * 2) = 1 -> the best-overlapping one comes back
    assert vfs.add_needed_vectors([], []) == []


def test_femininity_score_counts_distinct_segments():
    assert vfs.get_femininity_score([(0, 1, 0.6), (1, 2, 0.4), (2, 3, 0.5)]) == 2 / 3
    assert vfs.get_femininity_score([(0, 1, 0.6), (0, 1, 0.2)]) == 0.0       # same segment: the later prediction wins
    assert vfs.speech_intervals([('speech', 0, 1.5), ('music', 1.5, 2), ('speech', 2, 3)]) == [(0.0, 1.5), (2.0, 3.0)]


@pytest.mark.gpu
def test_pipeline_with_stand_in_weights():
    from oracle import vbx as ovbx
    from inaspeechsegmenter_amd.io import media2sig16kmono
    v = vfs.VoiceFemininityScoring(ffmpeg=None, models='synthetic')
    wav = os.path.join(GOLDEN, 'lamartine.wav')
    with pytest.raises(AssertionError):
        vfs.VoiceFemininityScoring(gd_model_criteria='xyz', ffmpeg=None, models='synthetic')
    v.vad = lambda path: [('noEnergy', 0.0, 0.5), ('speech', 0.5, 4.0), ('music', 4.0, 6.0), ('speech', 6.0, 14.0)]
    score, dur, nvec = v(wav)
    assert dur == 3.5 + 8.0 and 0.0 <= score <= 1.0 and nvec > 10
    # same result from the oracle's features + torch-CPU ResNet on the windows the VAD rule keeps
    sig = media2sig16kmono(wav, ffmpeg=None, dtype='float64')
    fea = ovbx.get_features(sig)
    speech = vfs.speech_intervals(v.vad(wav))
    wins = [(a, b) for a, b in ovbx.window_list(len(fea))]
    xs = [(f'w{a}', (round(a / 100.0, 3), round(b / 100.0, 3) if b - a == 144 else round(len(sig) / 16000, 3)), (a, b)) for a, b in wins]
    kept = vfs.apply_vad(xs, speech, v.vad_thresh)
    assert len(kept) == nvec
    emb = np.stack([ovbx.resnet101_forward(v.xvector_model.params, fea[a:b].T[None])[0] * 10 for _, _, (a, b) in kept[:6]])
    dev = np.stack([v.xvector_model.get_embedding(fea[a:b]) * 10 for _, _, (a, b) in kept[:6]])
    assert np.abs(dev - emb).max() <= 1e-3 * np.abs(emb).max()
    v.vad = lambda path: [('music', 0.0, 14.0)]
    assert v(wav) == (None, 0, 0)
