"""Voice femininity scoring on top of the MI355X kernels: host mirror of vbx_segmenter.VoiceFemininityScoring
(vbx_segmenter.py:92-202) -- SURVEY.md 8(f) item 3.

Pipeline (same order as the reference's __call__, :147-202): decode -> smn VAD (Segmenter, no gender) ->
get_features (csrc/vbx.hip) -> ResNet-101 x-vectors on every 144-frame window (engine, batched) -> keep the
x-vectors whose midpoint lies in speech and that overlap speech by >= vad_thresh (:129-145, 28-52) -> gender MLP
(a Keras Dense stack, run on the same engine) -> share of windows with p >= 0.5 (:55-61).

The reference does the interval arithmetic with pyannote.core (Annotation / Timeline.crop / Timeline.duration).  The helpers
below follow that package's rules on plain tuples -- a segment of <= 1e-6 s is empty, a timeline is a SET of segments,
crop() and duration() work on the SUPPORT (touching / overlapping segments merged), `intersects` wants more than 1e-6 s of
overlap -- and tests/test_vfs.py compares them with a class-by-class restatement of the package (oracle/pyannote_core.py)
running the reference's own statements, on random timelines incl. equal boundaries, overlaps, duplicates and empty
segments.  PARITY UNPINNED against the package itself (absent here) and against the only reference test of this tail
(run_test.py:177-187, score 0.534884 on lamartine.wav), which needs the un-vendored weights (final.onnx / raw_81.pth,
interspeech2023_*.hdf5).
Known divergence: add_needed_vectors (:40-52) reads `s.stop` on a pyannote Segment, which has no such attribute --
the branch would raise in the reference; here it does what the code evidently intends (append (key, (start, end), x)).
"""
import os

import numpy as np

from . import _native, keras_model
from .io import media2sig16kmono
from .segmenter import Segmenter, locate_model
from .vbx import FeatureExtractor, VBxExtractor, SR

_MLP_NET = 3          # engine net id of the gender MLP (0/1: VAD / gender CNNs, 4..: ResNet programs)


PRECISION = 1e-6     # pyannote.core's SEGMENT_PRECISION: a segment with end - start <= 1e-6 s is EMPTY (dropped / ignored)


def _nonempty(s, e):
    return (e - s) > PRECISION


def _intersects(a0, a1, b0, b1):
    """pyannote.core Segment.intersects: more than PRECISION of overlap, or equal starts."""
    return (a0 < b0 and b0 < a1 - PRECISION) or (a0 > b0 and a0 < b1 - PRECISION) or (a0 == b0)


def _support(intervals):
    """pyannote.core Timeline.support(): the distinct non-empty segments in (start, end) order, neighbours merged unless a gap
    of more than PRECISION separates them."""
    segs = sorted({(s, e) for s, e in intervals if _nonempty(s, e)})
    out = []
    for s, e in segs:
        if out and not _nonempty(min(e, out[-1][1]), max(s, out[-1][0])):       # `segment ^ new_segment` is empty: no gap
            out[-1] = (min(s, out[-1][0]), max(e, out[-1][1]))
        else:
            out.append((s, e))
    return out


def speech_intervals(vad_tuples):
    """get_annot_VAD (vbx_segmenter.py:64-69): the 'speech' segments as an Annotation keeps them -- one entry per distinct
    non-empty Segment(start, end), in (start, end) order."""
    return sorted({(s, e) for lab, s, e in vad_tuples if lab == 'speech' and _nonempty(s, e)})


def speech_duration(speech):
    """annot_vad.label_duration("speech") (vbx_segmenter.py:163): the duration of the SUPPORT of the speech timeline."""
    return sum(e - s for s, e in _support(speech))


def is_mid_speech(start, stop, speech):
    """vbx_segmenter.py:28-37: midpoint strictly inside a speech segment."""
    m = (start + stop) / 2
    return any(s < m < e for s, e in speech)


def cropped_duration(start, stop, speech):
    """Timeline([Segment(start, stop)]).crop(vad.get_timeline()).duration()  (vbx_segmenter.py:138-142; crop mode
    'intersection'): the support of the speech timeline is taken first, every support segment that `intersects` the window
    contributes window & segment unless that piece is empty, and the pieces' own support is summed in time order."""
    if not _nonempty(start, stop):
        return 0
    pieces = set()
    for s, e in _support(speech):
        if (s, e) <= (stop, stop) and _intersects(start, stop, s, e):
            p = (max(start, s), min(stop, e))
            if _nonempty(*p):
                pieces.add(p)
    return sum(e - s for s, e in _support(pieces))


def overlap_ratio(start, stop, speech):
    return cropped_duration(start, stop, speech) / (stop - start)


def add_needed_vectors(xvectors, t_mid):
    """vbx_segmenter.py:40-52: keep at least round(50 %) of the mid-in-speech predictions, best overlap first."""
    min_pred = round(0.5 * len(t_mid))
    if len(xvectors) < min_pred:
        order = np.argsort(np.asarray([t[0] for t in t_mid], dtype=np.float64))[::-1]
        ranked = [t_mid[i] for i in order]
        diff = min_pred - len(xvectors)
        for _, k, (s, e), x in ranked[len(xvectors):len(xvectors) + diff]:
            xvectors.append((k, (s, e), x))
    return xvectors


def apply_vad(xvectors, speech, vad_thresh):
    """vbx_segmenter.py:129-145."""
    midpoint_seg, kept = [], []
    for key, (start, stop), x in xvectors:
        if is_mid_speech(start, stop, speech):
            r = overlap_ratio(start, stop, speech)
            if r >= vad_thresh:
                kept.append((key, (start, stop), x))
            midpoint_seg.append((r, key, (start, stop), x))
    return add_needed_vectors(kept, midpoint_seg)


def get_femininity_score(g_preds):
    """vbx_segmenter.py:55-61: an Annotation keyed by Segment(start, stop) keeps ONE entry per distinct non-empty segment
    (a later prediction on the same segment replaces the earlier one); score = #(p >= 0.5) / #segments."""
    seen = {}
    for start, stop, p in g_preds:
        if _nonempty(start, stop):
            seen[(float(start), float(stop))] = bool(p >= 0.5)
    return sum(seen.values()) / len(seen)


def _load_resnet_params(path):
    """final.onnx -- what the reference's live backend loads (OnnxBackendExtractor, vbx_segmenter.py:249-266) -- through the
    package's own protobuf walk (onnx_reader.py); or raw_81.pth, the torch checkpoint behind the reference's commented
    TorchBackendExtractor (:268-288)."""
    if path.lower().endswith('.onnx'):
        from .onnx_reader import load_resnet101_params
        return load_resnet101_params(path)
    if path.lower().endswith('.npz'):                     # locate_model also accepts a flat '<stem>.npz' export of either file
        with np.load(path) as z:
            return {k: np.asarray(z[k]) for k in z.files}
    import torch
    ck = torch.load(path, map_location='cpu')
    sd = ck.get('state_dict', ck)
    return {k: v.numpy() for k, v in sd.items() if hasattr(v, 'numpy')}


class VoiceFemininityScoring:
    def __init__(self, gd_model_criteria='bgc', backend='onnx', ffmpeg='ffmpeg', device=0, models=None):
        """gd_model_criteria / backend: as vbx_segmenter.py:97-127.  models: None -> files from the
        remote_utils search path (final.onnx for the x-vector net, read by onnx_reader.py; raw_81.pth if only that exists);
        'synthetic' -> seeded stand-ins; or a dict {'resnet': state_dict-like, 'mlp': (layers, in_shape),
        'vad': the `models` argument of the inner Segmenter (optional; default = its Keras files)}."""
        assert backend in ['onnx'], "Backend should be 'onnx' (or 'pytorch' if uncommented)."
        assert gd_model_criteria in ['bgc', 'vfp'], "Gender detection model Criteria must be 'bgc' (default) or 'vfp'"
        gd_model, self.vad_thresh = ('interspeech2023_all.hdf5', 0.7) if gd_model_criteria == 'bgc' else ('interspeech2023_cvfr.hdf5', 0.62)
        self.ffmpeg = ffmpeg
        if models == 'synthetic':
            vad_models = 'synthetic'
        elif isinstance(models, dict):
            vad_models = models.get('vad')                # {'keras_speech_music_noise_cnn.hdf5': (layers, in_shape)} or 'synthetic'
        else:
            vad_models = None
        self.vad = Segmenter(vad_engine='smn', detect_gender=False, ffmpeg=ffmpeg, device=device, models=vad_models)
        self.ctx = self.vad.ctx
        if models == 'synthetic':
            rng = np.random.default_rng(23)
            resnet = keras_model.synthetic_resnet101()
            mlp = ([dict(type='dense', W=rng.normal(0, 0.1, (256, 64)).astype(np.float32), b=np.zeros(64, np.float32), activation='relu'),
                    dict(type='dense', W=rng.normal(0, 0.3, (64, 1)).astype(np.float32), b=np.zeros(1, np.float32), activation='sigmoid')],
                   (1, 1, 256))
        elif isinstance(models, dict):
            resnet, mlp = models['resnet'], models['mlp']
        else:
            try:                                          # the file `get_remote` fetches by default (remote_utils.py:13, backend='onnx')
                onnx_path = locate_model('final.onnx')
            except FileNotFoundError as exc:
                # no final.onnx: the torch checkpoint of the same network (remote_utils.py:14) if it is there, else that error
                try:
                    resnet = _load_resnet_params(locate_model('raw_81.pth'))
                except FileNotFoundError:
                    raise exc
            else:
                # a final.onnx the reader does not recognise as resnet.py's graph is an error of its own: no silent second try
                # (a torch ImportError from the .pth path would hide the real cause)
                try:
                    resnet = _load_resnet_params(onnx_path)
                except (ValueError, NotImplementedError) as exc:
                    raise type(exc)(f"{exc} -- {onnx_path} was found but could not be read as the ResNet-101 of resnet.py; remove it to "
                                    "fall back to raw_81.pth, or export it with `torch.onnx.export` from resnet.py") from exc
            mlp = keras_model.load_model_file(locate_model(gd_model))
        self.features = FeatureExtractor(self.ctx)
        self.xvector_model = VBxExtractor(self.ctx, resnet)
        layers, in_shape = mlp
        self.mlp_layers = layers                          # (kept for inspection / tests; the engine holds the compiled program)
        if len(in_shape) == 1:
            in_shape = (1, 1, in_shape[0])
        self.ctx.cnn_load(_MLP_NET, keras_model.compile_layers(layers, in_shape, patch_input=False))

    def gender_predict(self, x):
        x = np.asarray(x, dtype=np.float32)
        return self.ctx.cnn_forward(_MLP_NET, x.reshape(len(x), 1, 1, -1))

    def __call__(self, fpath):
        """-> (score, speech_duration, nb_vectors)  (vbx_segmenter.py:147-202)."""
        basename = os.path.splitext(os.path.basename(fpath))[0]
        signal = media2sig16kmono(fpath, ffmpeg=self.ffmpeg, dtype='float64')
        duration = len(signal) / SR
        speech = speech_intervals(self.vad(fpath))
        speech_dur = speech_duration(speech)
        if not speech_dur:
            return None, speech_dur, 0
        feats = self.features(signal)
        x_vectors = self.xvector_model(basename, feats, duration)
        x_vectors = apply_vad(x_vectors, speech, self.vad_thresh)
        if not x_vectors:                                 # (the reference fails inside the MLP predict on an empty batch)
            return None, speech_dur, 0
        pred = self.gender_predict(np.asarray([x for _, _, x in x_vectors])).reshape(len(x_vectors), -1)[:, 0]
        g = [(seg[0], seg[1], p) for (_, seg, _), p in zip(x_vectors, pred)]
        return get_femininity_score(g), speech_dur, len(g)
