"""GPU: the drop-in `Segmenter` end to end (decode -> device features -> CNNs -> compiled Viterbi
-> CSV/TextGrid) against the reference's weight-free goldens and against the oracle pipeline
running the same seeded stand-in weights."""
import filecmp
import os

import numpy as np
import pytest

from inaspeechsegmenter_amd import Segmenter, seg2csv
from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn
from conftest import GOLDEN, read_wav_int16, synth_pcm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def seg():
    return Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic')


def _oracle_segmentation(seg, pcm, vad='smn'):
    sig = (pcm / 32768.0).astype(np.float32)
    mspec, loge, difflen = osk.media2feats(sig)
    vp = lambda b: ocnn.forward(seg.vad.layers, b)
    gp = (lambda b: ocnn.forward(seg.gender.layers, b)) if seg.detect_gender else None
    return oseg.segment_feats(mspec, loge, difflen, 0, vad, vp, gp)


def _csv_rows(path):
    rows = [l.rstrip('\n').split('\t') for l in open(path)][1:]
    return [(r[0], float(r[1]), float(r[2])) for r in rows]


def test_silence_golden_csv_byte_identical(seg, tmp_path):
    out = tmp_path / 's.csv'
    t, nb, avg, lmsg = seg.batch_process([os.path.join(GOLDEN, 'silence2sec.wav')], [str(out)])
    assert nb == 1 and lmsg[0][1] == 0 and lmsg[0][2].startswith('ok ')
    assert filecmp.cmp(str(out), os.path.join(GOLDEN, 'silence2sec-smn-gender.csv'), shallow=False)


def test_musanmix_energy_rows_match_reference_golden(seg):
    res = seg(os.path.join(GOLDEN, 'musanmix.wav'))
    gold = _csv_rows(os.path.join(GOLDEN, 'musanmix-smn-gender.csv'))
    assert [(s, e) for l, s, e in res if l == 'noEnergy'] == [(s, e) for l, s, e in gold if l == 'noEnergy']
    for i in range(len(res) - 1):                               # run_test.py:68-88 test_boundaries
        assert res[i][2] == res[i + 1][1]
    assert res[0][1] == 0.0 and res[-1][2] == 74.5
    assert set(l for l, _, _ in res) <= {'noEnergy', 'music', 'noise', 'male', 'female'}


def test_musanmix_identical_to_oracle_pipeline(seg):
    pcm = read_wav_int16(os.path.join(GOLDEN, 'musanmix.wav'))
    assert seg.segment_signal(pcm) == _oracle_segmentation(seg, pcm)


def test_musanmix_reference_golden_csv_byte_identical(seg, tmp_path):
    """run_test.py:90-105 test_processingresult shape: the CSV written for media/musanmix.wav equals the reference's golden
    file byte for byte -- labels the NETWORKS decide (music, noise, noise, male, male, male) included.  The stand-in networks
    were calibrated on the CPU oracle to reproduce these labels (tests/golden/make_standin_heads.py); what this test adds is
    that the GPU engine makes the same decisions with them: a kernel bug that moved the probabilities would change rows."""
    out = tmp_path / 'musanmix.csv'
    seg2csv(seg(os.path.join(GOLDEN, 'musanmix.wav')), str(out))
    assert filecmp.cmp(str(out), os.path.join(GOLDEN, 'musanmix-smn-gender.csv'), shallow=False), open(out).read()


def test_networks_decide(seg):
    """Guard against a degenerate stand-in: on musanmix every VAD class wins > 10 % of the evaluated slots, on the
    generator's recording every class of both networks wins >= 15 % and >= 3 boundaries are CNN-driven; the GPU's arg-max
    equals the oracle's on every evaluated slot."""
    import bench
    from inaspeechsegmenter_amd import segmenter as S
    pcm = read_wav_int16(os.path.join(GOLDEN, 'musanmix.wav'))
    mspec, loge, difflen = osk.media2feats((pcm / 32768.0).astype(np.float32))
    l0 = oseg.energy_seglist(loge, 0.03)
    l1, raw = oseg.dnn_segment('smn', lambda b: ocnn.forward(seg.vad.layers, b), mspec, l0, difflen, return_raw=True)
    seg.segment_signal(pcm)                                                   # leaves musanmix's features resident
    idx = np.concatenate([np.arange(a, b) for lab, a, b in l0 if lab == 'energy'])
    p, fin = seg.ctx.cnn_probs(0, S._window_rows(len(loge))[idx])
    hist = np.bincount(p.argmax(1), minlength=3) / len(p)
    assert hist.min() > 0.10, hist
    srt = np.sort(raw, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 1e-3                                # (a near-tie may go either way within 1e-4)
    assert np.array_equal(p.argmax(1)[decided], raw.argmax(1)[decided]) and decided.mean() > 0.99
    # north star: frame logits within 1e-3 of the fp32 reference; log-probabilities are logits up to the common log-sum-exp
    assert np.abs(np.log(p) - np.log(raw)).max() < 1e-3 and np.abs(p - raw).max() < 2e-4
    pcm = bench.synth_recording(0, 200 * 16000, 'cpu').numpy()
    res = seg.segment_signal(pcm)
    assert res == _oracle_segmentation(seg, pcm)
    dur = {}
    for lab, a, b in res:
        dur[lab] = dur.get(lab, 0.0) + b - a
    vad_total = sum(v for k, v in dur.items() if k != 'noEnergy')
    assert min(dur['music'], dur['noise'], dur['male'] + dur['female']) / vad_total >= 0.15, dur
    assert min(dur['male'], dur['female']) / (dur['male'] + dur['female']) >= 0.15, dur
    assert bench.cnn_driven_boundaries(res) >= 3


def test_synthetic_signals_identical_to_oracle_pipeline(seg):
    for seed, n in ((1, 160000), (2, 48000), (3, 400 + 160 * 67 + 5)):
        pcm = synth_pcm(seed, n)
        assert seg.segment_signal(pcm) == _oracle_segmentation(seg, pcm), seed


def test_short_media_path(seg):
    pcm = synth_pcm(77, 16000)[5000:5000 + 400 + 160 * 65]       # 66 frames, like media/0021.mp3
    with pytest.warns(UserWarning, match='duration is short'):
        res = seg.segment_signal(pcm)
    assert res[-1][2] == 0.66
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        assert res == _oracle_segmentation(seg, pcm)


def test_sm_engine_and_no_gender():
    s2 = Segmenter(vad_engine='sm', detect_gender=False, ffmpeg=None, models='synthetic')
    pcm = synth_pcm(9, 96000)
    res = s2.segment_signal(pcm)
    assert res == _oracle_segmentation(s2, pcm, 'sm')
    assert set(l for l, _, _ in res) <= {'noEnergy', 'speech', 'music'}


def test_sm_engine_reference_golden(seg):
    """media/musanmix-sm-gender.csv: same label sequence, every boundary within 0.2 s (its CNN-driven boundary at 32.48 s --
    `male` | `music` inside one energy segment -- is where the Viterbi path of the fitted stand-in switches), and the GPU
    result equals the oracle pipeline exactly."""
    s2 = Segmenter(vad_engine='sm', detect_gender=True, ffmpeg=None, models='synthetic')
    res = s2(os.path.join(GOLDEN, 'musanmix.wav'))
    gold = _csv_rows(os.path.join(GOLDEN, 'musanmix-sm-gender.csv'))
    assert [r[0] for r in res] == [g[0] for g in gold]
    assert max(max(abs(a - c), abs(b - d)) for (_, a, b), (_, c, d) in zip(res, gold)) <= 0.2
    assert res == _oracle_segmentation(s2, read_wav_int16(os.path.join(GOLDEN, 'musanmix.wav')), 'sm')


def test_batch_process_contract(seg, tmp_path):
    src = os.path.join(GOLDEN, 'musanmix.wav')
    lout = [str(tmp_path / 'a' / '1.csv'), str(tmp_path / '2.csv'), str(tmp_path / '3.csv'), str(tmp_path / '4.TextGrid')]
    t, nb, avg, lmsg = seg.batch_process([src, src, os.path.join(GOLDEN, 'doesnotexist.wav')], lout[:3])
    assert nb == 2 and [m[1] for m in lmsg] == [0, 0, 2] and lmsg[2][2].startswith('error: ')
    assert filecmp.cmp(lout[0], lout[1], shallow=False)          # run_test.py:107-120
    ref = tmp_path / 'ref.csv'
    seg2csv(seg(src), str(ref))
    assert filecmp.cmp(lout[0], str(ref), shallow=False)
    t, nb, avg, lmsg = seg.batch_process([src, src], lout[:2], skipifexist=True)
    assert nb == 0 and avg == -1 and [m[1:] for m in lmsg] == [(1, 'already exists')] * 2
    t, nb, avg, lmsg = seg.batch_process([src], [lout[3]], output_format='textgrid')
    assert nb == 1 and open(lout[3]).read().startswith('File type = "ooTextFile"')
    with pytest.raises(NotImplementedError):
        seg.batch_process([src], [lout[0]], output_format='json')


def test_pipeline_workers_follow_the_segmenters_settings(tmp_path):
    """The extra device contexts of the multi-file pipeline take over the arithmetic mode and workspace cap of seg.ctx, at
    creation and again on every call (they are cached): f32 batch_process with two workers == per-file f32 calls."""
    from inaspeechsegmenter_amd import _native
    s2 = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models='synthetic')
    lin = []
    for i in range(4):
        _wav(tmp_path / f'p{i}.wav', synth_pcm(40 + i, 16000 * 20 + 7 * i))
        lin.append(str(tmp_path / f'p{i}.wav'))
    for prec, wsl in ((_native.PREC_F32, 1 << 30), (_native.PREC_BF16X3, 2 << 30)):
        s2.ctx.set_precision(prec)
        s2.ctx.set_workspace_limit(wsl)
        lout = [str(tmp_path / f'o{prec}_{i}.csv') for i in range(4)]
        t, nb, avg, lmsg = s2.batch_process(lin, lout, batch_files=1, workers=2)
        assert nb == 4
        ws = s2.__dict__['_pipeline_workers']
        assert len(ws) == 2 and all(w.ctx.precision == prec and w.ctx.workspace_limit == wsl for w in ws)
        for src, dst in zip(lin, lout):
            ref = str(tmp_path / 'ref.csv')
            seg2csv(s2(src), ref)
            assert filecmp.cmp(dst, ref, shallow=False)
        secs = [float(m[2].split()[1]) for m in lmsg]
        assert all(0.0 < v < t for v in secs) and sum(secs) < 2.5 * t          # per-file figures, not time since batch start
    s2.close()


def test_constructor_contract():
    with pytest.raises(Exception, match='ffmpeg program not found'):
        Segmenter(ffmpeg='definitely-not-ffmpeg', models='synthetic')
    with pytest.raises(AssertionError):
        Segmenter(vad_engine='xyz', ffmpeg=None, models='synthetic')
    with pytest.raises(FileNotFoundError):
        Segmenter(ffmpeg=None)                                   # real weights absent on this box


def test_segment_feats_accepts_host_arrays(seg):
    g = np.load(os.path.join(GOLDEN, 'sidekit_feats.npz'))
    a = seg.segment_feats(g['synth_mspec'], g['synth_loge'], 0, 1.5)
    pcm = synth_pcm(1234, 48000)
    b = seg.segment_signal(pcm, start_sec=1.5)
    assert a == b


def test_cli_end_to_end(tmp_path):
    """run_test.py:136-148 test_program / test_program_smn shape: the CLI writes <basename>.csv per input; the
    weight-free golden (silence2sec) is byte-identical."""
    import subprocess
    import sys
    root = os.path.dirname(GOLDEN.rstrip('/').rsplit('/tests', 1)[0] + '/x')
    cli = os.path.join(os.path.dirname(os.path.dirname(GOLDEN)), 'scripts', 'ina_speech_segmenter_amd.py')
    r = subprocess.run([sys.executable, cli, '-i', os.path.join(GOLDEN, 'silence2sec.wav'), os.path.join(GOLDEN, 'musanmix.wav'),
                        '-o', str(tmp_path), '-b', 'None', '--models', 'synthetic'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert filecmp.cmp(str(tmp_path / 'silence2sec.csv'), os.path.join(GOLDEN, 'silence2sec-smn-gender.csv'), shallow=False)
    rows = _csv_rows(str(tmp_path / 'musanmix.csv'))
    gold = _csv_rows(os.path.join(GOLDEN, 'musanmix-smn-gender.csv'))
    assert [(s, e) for l, s, e in rows if l == 'noEnergy'] == [(s, e) for l, s, e in gold if l == 'noEnergy']


def test_batch_process_many_files_equals_single_calls(seg, tmp_path):
    """configs[2] shape at test scale: several 1-minute WAV files through batch_process (decode thread, H2D,
    device, Viterbi, CSV) give byte-identical CSVs to per-file __call__ + seg2csv."""
    import struct
    lin, lout = [], []
    for i in range(5):
        pcm = synth_pcm(100 + i, 16000 * 60 + 37 * i)
        p = tmp_path / f'in{i}.wav'
        with open(p, 'wb') as f:
            f.write(b'RIFF' + struct.pack('<I', 36 + pcm.nbytes) + b'WAVEfmt ' + struct.pack('<IHHIIHH', 16, 1, 1, 16000, 32000, 2, 16)
                    + b'data' + struct.pack('<I', pcm.nbytes) + pcm.tobytes())
        lin.append(str(p))
        lout.append(str(tmp_path / f'out{i}.csv'))
    t, nb, avg, lmsg = seg.batch_process(lin, lout)
    assert nb == 5 and all(m[1] == 0 for m in lmsg)
    for src, dst in zip(lin, lout):
        ref = str(tmp_path / 'ref.csv')
        seg2csv(seg(src), ref)
        assert filecmp.cmp(dst, ref, shallow=False), src


def test_device_resident_pcm_entry_equals_host_path(seg):
    """bench.py's entry (PCM16 already in HBM, slot-unit result, dense evaluation) == the host-signal path."""
    import torch
    pcm = synth_pcm(321, 16000 * 90 + 11)
    t = torch.from_numpy(pcm).cuda()
    torch.cuda.synchronize()
    slots_dense = seg.segment_device_pcm(t.data_ptr(), t.numel(), dense=True)
    slots_ref = seg.segment_device_pcm(t.data_ptr(), t.numel(), dense=False)
    host = seg.segment_signal(pcm)
    assert slots_dense == slots_ref
    assert [(l, a * .02, b * .02) for l, a, b in slots_ref] == host
    with pytest.raises(ValueError):
        seg.segment_device_pcm(t.data_ptr(), 400 + 160 * 60)          # < 68 frames: use segment_signal


def _wav(path, data, fmt_tag=1, bits=16):
    import struct
    raw = data.tobytes()
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', 36 + len(raw)) + b'WAVEfmt ' +
                struct.pack('<IHHIIHH', 16, fmt_tag, 1, 16000, 16000 * bits // 8, bits // 8, bits) + b'data' + struct.pack('<I', len(raw)) + raw)


def test_pipeline_mixed_batch_equals_single_calls(seg, tmp_path):
    """The super-batch path of pipeline.process_files: files of very different lengths (none a multiple of 160 samples)
    packed into one device pass, several passes per call (batch_files=3), a 0.5 s file (< 68 frames: single-file path with
    the mel padding of segmenter.py:61-65), a float32 WAV (single-file path), silence, an undecodable and a missing file --
    every CSV byte-identical to the per-file call, errors reported per file, with one and with two device workers."""
    specs = [('a', synth_pcm(11, 16000 * 31 + 77)), ('b', synth_pcm(12, 16000 * 7 + 3)), ('short', synth_pcm(13, 16000)[4000:12000]),
             ('c', synth_pcm(14, 16000 * 95 + 159)), ('sil', np.zeros(16000 * 4 + 5, np.int16)), ('d', synth_pcm(15, 16000 * 12 + 1)),
             ('e', synth_pcm(16, 16000 * 3 + 80))]
    lin = []
    for name, pcm in specs:
        _wav(tmp_path / f'{name}.wav', pcm)
        lin.append(str(tmp_path / f'{name}.wav'))
    fl = (synth_pcm(17, 16000 * 9 + 41) / 32768.0).astype('<f4')
    _wav(tmp_path / 'float.wav', fl, fmt_tag=3, bits=32)
    lin.insert(3, str(tmp_path / 'float.wav'))
    (tmp_path / 'junk.wav').write_bytes(b'not a wav file at all')
    lin.insert(5, str(tmp_path / 'junk.wav'))
    lin.append(str(tmp_path / 'missing.wav'))
    for workers in (1, 2):
        lout = [str(tmp_path / f'out{workers}' / (os.path.basename(p)[:-4] + '.csv')) for p in lin]
        t, nb, avg, lmsg = seg.batch_process(lin, lout, batch_files=3, workers=workers)
        codes = [m[1] for m in lmsg]
        assert [m[0] for m in lmsg] == lout                                   # input order
        assert codes == [0, 0, 0, 0, 0, 2, 0, 0, 0, 2] and nb == 8, lmsg
        assert all(m[2].startswith('error: ') for m in lmsg if m[1] == 2)
        for src, dst, code in zip(lin, lout, codes):
            if code == 0:
                ref = str(tmp_path / 'ref.csv')
                with __import__('warnings').catch_warnings():
                    __import__('warnings').simplefilter('ignore')
                    seg2csv(seg(src), ref)
                assert filecmp.cmp(dst, ref, shallow=False), (workers, src)
    # and the archive driver on the same list (single process: the table holds every good file)
    from inaspeechsegmenter_amd import archive
    table, lmsg = archive.segment_archive(seg, lin, None)
    assert sorted(table) == [0, 1, 2, 3, 4, 6, 7, 8] and [m[1] for m in lmsg] == codes
    with __import__('warnings').catch_warnings():
        __import__('warnings').simplefilter('ignore')
        assert table[2] == seg(lin[2]) and table[6] == seg(lin[6])


@pytest.mark.parametrize('topology', ['conv1_same', 'conv2_7x7', 'relu_then_bn'])
def test_other_topologies_through_the_file_pipeline(tmp_path, topology):
    """A Segmenter whose two nets are NOT the stand-in topology -- a zero-padded first conv (FS form: per-window edge rows indexed by the
    launch-local window), a 7x7 second conv (ring form, rows per tile < 512), BatchNorm behind the activation (folded forward by the
    lowering) -- through batch_process (several files laid end to end in one super-batch, window lists offset per file) must give
    what the per-file calls give, dense batches included, and what the oracle pipeline gives on the same layers."""
    import topologies as TP
    from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn
    nets = TP.nets(topology, seed=3)
    models = {'keras_speech_music_noise_cnn.hdf5': nets['vad'], 'keras_male_female_cnn.hdf5': nets['gender']}
    s2 = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models=models)
    lin = []
    for i in range(5):
        _wav(tmp_path / f't{i}.wav', synth_pcm(70 + i, 16000 * (12 + 5 * i) + 13 * i))
        lin.append(str(tmp_path / f't{i}.wav'))
    single = [s2(p) for p in lin]
    for dense in (False, True):
        s2.dense_batches = dense
        lout = [str(tmp_path / f'o{int(dense)}_{i}.csv') for i in range(5)]
        t, nb, avg, lmsg = s2.batch_process(lin, lout, batch_files=3, workers=2)
        assert nb == 5, lmsg
        for seg_i, dst in zip(single, lout):
            ref = str(tmp_path / 'ref.csv')
            seg2csv(seg_i, ref)
            assert filecmp.cmp(dst, ref, shallow=False), (topology, dense, dst)
    s2.dense_batches = False
    # the oracle pipeline on file 2 (random weights: the labels mean nothing, the identity of the segmentation does)
    from inaspeechsegmenter_amd.io import decode_pcm
    pcm = decode_pcm(lin[2], None, None, None)
    mspec, loge, difflen = osk.media2feats((pcm / 32768.0).astype(np.float32))
    want = oseg.segment_feats(mspec, loge, difflen, 0, 'smn', lambda b: ocnn.forward(nets['vad'][0], b), lambda b: ocnn.forward(nets['gender'][0], b))
    got = single[2]
    assert [g[0] for g in got] == [w[0] for w in want] and [g[1:] for g in got] == [w[1:] for w in want], (got, want)
    s2.close()


def test_segmenter_loads_keras_hdf5_files_from_the_model_dir(tmp_path, monkeypatch):
    """The reference's default construction -- `Segmenter()` finds `keras_speech_music_noise_cnn.hdf5` / `keras_male_female_cnn.hdf5` in
    the Keras cache directory (remote_utils.py:18-27, segmenter.py:129-131) -- with two small Keras-layout HDF5 files written by the real
    h5py (tests/golden/make_keras_hdf5.py) staged under those names: found by locate_model, read by the package's own HDF5 reader (no
    h5py in this image), lowered, loaded on the device, and the segmentation is what the oracle pipeline gives on the same layers."""
    import shutil
    from inaspeechsegmenter_amd import segmenter as S, keras_model as KM
    d = tmp_path / 'keras_cache'
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, 'keras2_like.hdf5'), d / 'keras_speech_music_noise_cnn.hdf5')
    shutil.copy(os.path.join(GOLDEN, 'keras2_like_gender.hdf5'), d / 'keras_male_female_cnn.hdf5')
    monkeypatch.setattr(S, '_MODEL_DIRS', [str(d)])
    s2 = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None)          # no `models=`: the files decide
    vad_layers, shp = KM.load_model_file(str(d / 'keras_speech_music_noise_cnn.hdf5'))
    gen_layers, shp2 = KM.load_model_file(str(d / 'keras_male_female_cnn.hdf5'))
    assert shp == (68, 21, 1) and shp2 == (68, 24, 1) and len(s2.vad.layers) == len(vad_layers)
    pcm = synth_pcm(91, 16000 * 25 + 77)
    _wav(tmp_path / 'x.wav', pcm)
    got = s2(str(tmp_path / 'x.wav'))
    mspec, loge, difflen = osk.media2feats((pcm / 32768.0).astype(np.float32))
    want = oseg.segment_feats(mspec, loge, difflen, 0, 'smn', lambda b: ocnn.forward(vad_layers, b), lambda b: ocnn.forward(gen_layers, b))
    assert [g[0] for g in got] == [w[0] for w in want] and [g[1:] for g in got] == [w[1:] for w in want], (got, want)
    with pytest.raises(FileNotFoundError):                                     # the sm engine's file is not there: the reference's message
        Segmenter(vad_engine='sm', detect_gender=False, ffmpeg=None)
    s2.close()


def test_graph_shaped_models_through_the_file_pipeline(tmp_path):
    """A Segmenter whose two nets are functional models with branches (tests/graph_nets.py: a residual block in the stand-in's trunk,
    an inception-style concatenation) -- lowered as chains joined by ISS_OP_ELT rows -- through single calls and batch_process, against
    the oracle pipeline on the same layers: the product path computes whatever `keras.models.load_model` (segmenter.py:129-131) would
    have handed the reference."""
    import graph_nets as GN
    from oracle import sidekit as osk, segment as oseg, keras_cnn as ocnn
    vad, gender = GN.NETS['standin_residual'](21, 3, 3), GN.NETS['inception'](24, 2, 4)
    models = {'keras_speech_music_noise_cnn.hdf5': vad, 'keras_male_female_cnn.hdf5': gender}
    s2 = Segmenter(vad_engine='smn', detect_gender=True, ffmpeg=None, models=models)
    lin = []
    for i in range(3):
        _wav(tmp_path / f'g{i}.wav', synth_pcm(170 + i, 16000 * (14 + 6 * i) + 7 * i))
        lin.append(str(tmp_path / f'g{i}.wav'))
    single = [s2(p) for p in lin]
    lout = [str(tmp_path / f'go_{i}.csv') for i in range(3)]
    t, nb, avg, lmsg = s2.batch_process(lin, lout, batch_files=2, workers=2)
    assert nb == 3, lmsg
    for seg_i, dst in zip(single, lout):
        ref = str(tmp_path / 'gref.csv')
        seg2csv(seg_i, ref)
        assert filecmp.cmp(dst, ref, shallow=False), dst
    from inaspeechsegmenter_amd.io import decode_pcm
    pcm = decode_pcm(lin[1], None, None, None)
    mspec, loge, difflen = osk.media2feats((pcm / 32768.0).astype(np.float32))
    want = oseg.segment_feats(mspec, loge, difflen, 0, 'smn', lambda b: ocnn.forward(vad[0], b), lambda b: ocnn.forward(gender[0], b))
    got = single[1]
    assert [g[0] for g in got] == [w[0] for w in want] and [g[1:] for g in got] == [w[1:] for w in want], (got, want)
    s2.close()
