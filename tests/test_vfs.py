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


def _random_timeline(rng, mode):
    """speech tuples as a Segmenter-like source could hand them over, and nastier: grid-aligned cuts so that equal boundaries,
    touching, overlapping, duplicate and empty segments all occur."""
    grid = {'grid': 0.02, 'fine': 0.001, 'float': None}[mode]
    n = int(rng.integers(0, 9))
    out = []
    for _ in range(n):
        if grid:
            a = float(rng.integers(0, round(8 / grid))) * grid
            d = float(rng.integers(0, round(2.4 / grid))) * grid
        else:
            a = float(rng.uniform(0, 8))
            d = float(rng.choice([0.0, 1e-7, 9e-7, 1.1e-6, rng.uniform(0, 2.5)]))
        out.append(('speech', a, a + d))
        if rng.random() < 0.2:
            out.append(out[-1])                                       # duplicate
        if rng.random() < 0.25:
            out.append(('speech', a + d, a + d + float(rng.integers(1, 50)) * (grid or 0.013)))      # touching neighbour
        if rng.random() < 0.2:
            out.append(('music', a, a + d))
    rng.shuffle(out)
    return [tuple(t) for t in out]


@pytest.mark.parametrize('mode', ['grid', 'fine', 'float'])
def test_interval_helpers_follow_pyannote_core_semantics(mode):
    """vfs.py's tuple helpers == the reference's own statements (vbx_segmenter.py:28-69,129-145) executed on a class-by-class
    restatement of pyannote.core (oracle/pyannote_core.py): speech duration, the mid-point rule, the cropped duration, the kept
    x-vectors and the femininity score, on random timelines with equal boundaries / overlaps / duplicates / empty segments."""
    from oracle import pyannote_core as pc
    rng = np.random.default_rng({'grid': 11, 'fine': 12, 'float': 13}[mode])
    x = np.zeros(3)
    n_fallback = n_kept = n_merge = 0
    for it in range(400):
        vad = _random_timeline(rng, mode)
        a_vad = pc.get_annot_VAD(vad)
        speech = vfs.speech_intervals(vad)
        assert speech == [(s.start, s.end) for s, _, _ in a_vad.itertracks(yield_label=True)]
        assert vfs.speech_duration(speech) == a_vad.label_duration('speech')
        n_merge += len(a_vad.get_timeline().support()) < len(a_vad)
        # x-vector windows as VBxExtractor lays them out (0.24 s hop, 1.44 s long, :233-246) plus a ragged tail and, in float
        # mode, arbitrary ones
        wins = [(f'w{i}', (round(i * 0.24, 2), round(i * 0.24 + 1.44, 2)), x) for i in range(int(rng.integers(0, 40)))]
        if wins:
            last = wins[-1][1][0] + 0.24
            wins.append(('tail', (last, last + float(rng.integers(10, 143)) / 100.0), x))
        if mode == 'float':
            wins += [(f'r{i}', (float(a), float(a + rng.choice([0.0, 5e-7, rng.uniform(0.01, 2)]))), x) for i, a in enumerate(rng.uniform(0, 9, 6))]
        for _, (a, b), _ in wins:
            assert vfs.is_mid_speech(a, b, speech) == bool(pc.is_mid_speech(a, b, a_vad))
            want = pc.Timeline([pc.Segment(a, b)]).crop(a_vad.get_timeline()).duration()
            assert vfs.cropped_duration(a, b, speech) == want, (a, b, speech)
        wins = [w for w in wins if w[1][1] > w[1][0]]                    # (a zero-length window divides by zero in both)
        thr = float(rng.choice([0.62, 0.7]))
        try:
            want = pc.apply_vad(list(wins), a_vad, thr)                  # the reference's own statements ...
        except AttributeError as exc:                                    # ... whose fallback branch reads Segment.stop (:50)
            assert 'stop' in str(exc)
            n_fallback += 1
            want = pc.apply_vad(list(wins), a_vad, thr, fixed_stop_attribute=True)
        got = vfs.apply_vad(list(wins), speech, thr)
        tied = len({vfs.overlap_ratio(a, b, speech) for _, (a, b), _ in wins}) < len(wins)
        if not tied or len(got) == len([w for w in wins if vfs.is_mid_speech(*w[1], speech) and vfs.overlap_ratio(*w[1], speech) >= thr]):
            assert [(k, se) for k, se, _ in got] == [(k, se) for k, se, _ in want]
        else:                                                            # equal ratios: the fallback's order among them is an
            assert len(got) == len(want)                                 # unstable sort in the reference; the COUNT is defined
        n_kept += len(got)
        preds = [(a, b, float(rng.random())) for _, (a, b), _ in got]
        if preds and rng.random() < 0.3:
            preds.append((preds[0][0], preds[0][1], 1.0 - preds[0][2]))  # the same segment again: the later prediction wins
        if mode == 'float' and preds:
            preds.append((1.0, 1.0 + 5e-7, 0.9))                         # an empty segment: ignored by the Annotation
        if preds:
            def outcome(f):
                try:
                    return f(np.asarray(preds))
                except ZeroDivisionError:                                 # only empty segments: len(annotation) == 0 in both
                    return 'div0'
            assert outcome(vfs.get_femininity_score) == outcome(pc.get_femininity_score)
    assert n_kept > 100 and n_fallback > 3 and n_merge > 20, (n_kept, n_fallback, n_merge)


def test_pyannote_restatement_known_answers():
    """The documented behaviours the numbers hinge on (pyannote.core's own docstring examples and definitions)."""
    from oracle.pyannote_core import Annotation, Segment, Timeline
    assert not Segment(1.0, 1.0 + 1e-6) and Segment(1.0, 1.0 + 2e-6) and Segment(3, 2).duration == 0.
    assert (Segment(0, 10) & Segment(5, 15)) == Segment(5, 10) and not (Segment(0, 10) & Segment(15, 20))
    assert (Segment(0, 10) ^ Segment(15, 20)) == Segment(10, 15) and (Segment(0, 10) | Segment(5, 15)) == Segment(0, 15)
    assert Segment(0, 10).intersects(Segment(5, 15)) and not Segment(0, 10).intersects(Segment(10, 15))
    t = Timeline([Segment(0, 5), Segment(4, 6), Segment(6, 7), Segment(8, 9), Segment(8, 9), Segment(2, 2)])
    assert len(t) == 4 and list(t.support()) == [Segment(0, 7), Segment(8, 9)] and t.duration() == 8
    c = Timeline([Segment(3, 8.5)]).crop(t)
    assert list(c) == [Segment(3, 7), Segment(8, 8.5)] and c.duration() == 4.5
    a = Annotation()
    a[Segment(0, 1), '_'] = True
    a[Segment(0, 1), '_'] = False
    a[Segment(1, 2), '_'] = True
    a[Segment(5, 5), '_'] = True
    assert len(a) == 2 and len(a.label_timeline(True)) == 1 and len(a.label_timeline('x')) == 0
    with pytest.raises(AttributeError):
        Segment(0, 1).stop


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
    # the gender MLP (vbx_segmenter.py:189): the device's forward of the Dense stack == the oracle's Keras-semantics forward,
    # on the device's own x-vectors of every kept window; and the score is what the reference's statements give on them
    from oracle import keras_cnn as ocnn, pyannote_core as pc
    feats = v.features(sig)
    allx = v.xvector_model('lamartine', feats, len(sig) / 16000)
    a_vad = pc.get_annot_VAD(v.vad(wav))
    kept_ref = pc.apply_vad(list(allx), a_vad, v.vad_thresh)               # pyannote-semantics restatement of :129-145
    kept_dev = vfs.apply_vad(list(allx), speech, v.vad_thresh)
    assert [(k, se) for k, se, _ in kept_dev] == [(k, se) for k, se, _ in kept_ref] and len(kept_ref) == nvec
    X = np.asarray([x for _, _, x in kept_ref], np.float32)
    p_dev = v.gender_predict(X).reshape(len(X), -1)[:, 0]
    p_ora = ocnn.forward([dict(type='flatten')] + list(v.mlp_layers), X.reshape(len(X), 1, 1, -1)).reshape(len(X), -1)[:, 0]
    assert np.abs(p_dev - p_ora).max() <= 1e-4                               # probabilities; north star: logits within 1e-3
    g = np.asarray([(se[0], se[1], p) for (_, se, _), p in zip(kept_ref, p_ora)])
    assert score == pc.get_femininity_score(g) or np.abs(p_ora - 0.5).min() < 2e-4
    v.vad = lambda path: [('music', 0.0, 14.0)]
    assert v(wav) == (None, 0, 0)
