"""Oracle: segmentation bookkeeping around the two CNNs.  Test infrastructure only.

Restates the functions of /root/reference/inaSpeechSegmenter/segmenter.py that
cannot be imported here (the module pulls in tensorflow and skimage):
_energy_activity 69-73, _get_patches 76-88, _binidx2seglist 91-108,
DnnSegmenter.__call__ 135-179, Segmenter.segment_feats 250-276.
The network itself is a callable `predict(batch(N,68,h,1) f32) -> (N,C) f32`.

Pinned: tests/golden/ref_segmenter_pin.py executes the reference's own lines (ast-extracted, under the interpreter that
has skimage) on the committed feature fixtures with a bit-reproducible stand-in predictor; make_golden.py and
tests/test_oracle_golden.py assert that this file reproduces its patches bit for bit and its segments exactly
(tests/golden/segmenter_pin.npz).
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

from .viterbi import viterbi_decoding, pred2logemission, log_trans_exp, diag_trans_exp

# (outlabels, inlabel, nmel, viterbi_arg)  segmenter.py:182-204
NETS = {
    'sm': (('speech', 'music'), 'energy', 21, 150),
    'smn': (('speech', 'music', 'noise'), 'energy', 21, 80),
    'gender': (('female', 'male'), 'speech', 24, 80),
}


def energy_activity(loge, ratio):
    # segmenter.py:69-73
    with np.errstate(invalid='ignore'):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            threshold = np.mean(loge[np.isfinite(loge)]) + np.log(ratio)
    raw_activity = (loge > threshold)
    return viterbi_decoding(pred2logemission(raw_activity), log_trans_exp(150, cost0=-5))


def get_patches(mspec, w=68, step=2):
    """segmenter.py:76-88; skimage view_as_windows(mspec, (w,h), step) ==
    sliding_window_view(mspec, (w,h))[::step, 0]."""
    h = mspec.shape[1]
    data = sliding_window_view(mspec, (w, h))[::step, 0].reshape(-1, w * h)
    with np.errstate(invalid='ignore', divide='ignore'):
        data = (data - np.mean(data, axis=1).reshape((len(data), 1))) / np.std(data, axis=1).reshape((len(data), 1))
    lfill = [data[0, :].reshape(1, h * w)] * (w // (2 * step))
    rfill = [data[-1, :].reshape(1, h * w)] * (w // (2 * step) - 1 + len(mspec) % 2)
    data = np.vstack(lfill + [data] + rfill)
    finite = np.all(np.isfinite(data), axis=1)
    return data.reshape(len(data), w, h), finite


def binidx2seglist(binidx):
    # segmenter.py:91-108
    cur, beg, out = None, -1, []
    i = -1
    for i, e in enumerate(binidx):
        if e != cur:
            if cur is not None:
                out.append((cur, beg, i))
            cur, beg = e, i
    out.append((cur, beg, i + 1))
    return out


def dnn_segment(net, predict, mspec, lseg, difflen=0, return_raw=False):
    """DnnSegmenter.__call__, segmenter.py:135-179."""
    outlabels, inlabel, nmel, viterbi_arg = NETS[net]
    if nmel < 24:
        mspec = mspec[:, :nmel].copy()
    patches, finite = get_patches(mspec, 68, 2)
    if difflen > 0:
        patches = patches[:-int(difflen / 2), :, :]
        finite = finite[:-int(difflen / 2)]
    batch = [patches[start:stop, :] for lab, start, stop in lseg if lab == inlabel]
    raw_all = None
    if len(batch) > 0:
        batch = np.expand_dims(np.concatenate(batch), 3)
        rawpred = np.array(predict(batch.astype(np.float32)))
        raw_all = rawpred.copy()
    ret = []
    for lab, start, stop in lseg:
        if lab != inlabel:
            ret.append((lab, start, stop))
            continue
        n = stop - start
        r = rawpred[:n]
        rawpred = rawpred[n:]
        r[finite[start:stop] == False, :] = 0.5
        pred = viterbi_decoding(np.log(r), diag_trans_exp(viterbi_arg, len(outlabels)))
        for lab2, start2, stop2 in binidx2seglist(pred):
            ret.append((outlabels[int(lab2)], start2 + start, stop2 + start))
    if return_raw:
        return ret, raw_all
    return ret


def energy_seglist(loge, ratio=0.03):
    # segmenter.py:261-267
    lseg = []
    for lab, start, stop in binidx2seglist(energy_activity(loge, ratio)[::2]):
        lseg.append(('noEnergy' if lab == 0 else 'energy', start, stop))
    return lseg


def segment_feats(mspec, loge, difflen, start_sec, vad_net, vad_predict,
                  gender_predict=None, energy_ratio=0.03):
    # segmenter.py:250-276
    lseg = energy_seglist(loge, energy_ratio)
    lseg = dnn_segment(vad_net, vad_predict, mspec, lseg, difflen)
    if gender_predict is not None:
        lseg = dnn_segment('gender', gender_predict, mspec, lseg, difflen)
    return [(lab, start_sec + start * .02, start_sec + stop * .02) for lab, start, stop in lseg]
