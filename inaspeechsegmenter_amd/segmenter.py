"""`Segmenter`: drop-in host mirror of the reference's public class, driving the gfx950 kernels.

Same constructor kwargs, methods, return values and error behaviour as
/root/reference/inaSpeechSegmenter/segmenter.py (Segmenter 207-335, DnnSegmenter 111-204,
medialist2feats 338-374); the body is re-organised around the native pipeline:

    decode (host) -> iss_signal_* -> iss_sidekit  (features stay in HBM)
                  -> log-energy to host -> energy Viterbi (compiled host code)
                  -> iss_cnn_probs(VAD net, 'energy' slots)   -> Viterbi per segment
                  -> iss_cnn_probs(gender net, 'speech' slots) -> Viterbi per segment
                  -> [(label, start_sec + i*.02, ...)]

The 68-frame patches of `_get_patches` (segmenter.py:76-88) are never materialised: the
host only builds the int32 list "first mel row of the window feeding slot i".
"""
import os
import time
import shutil
import warnings

import threading
import numpy as np

from . import _native
from . import tables
from . import keras_model
from .io import decode_pcm
from .export_funcs import seg2csv, seg2textgrid

_MODEL_DIRS = ('/root/.keras/inaSpeechSegmenter/', os.path.expanduser('~/.keras/inaSpeechSegmenter/'))


# ---------------------------------------------------------------- Viterbi helpers (host)
def pred2logemission(pred, eps=1e-10):
    """viterbi_utils.py:29-34."""
    pred = np.asarray(pred)
    ret = np.full((len(pred), 2), eps)
    ret[pred == 0, 0] = 1 - eps
    ret[pred == 1, 1] = 1 - eps
    return np.log(ret)


def log_trans_exp(exp, cost0=0, cost1=0):
    """viterbi_utils.py:36-42."""
    ret = np.full((2, 2), -exp * np.log(10))
    ret[0, 0] = cost0
    ret[1, 1] = cost1
    return ret


def diag_trans_exp(exp, dim):
    """viterbi_utils.py:44-49."""
    ret = np.full((dim, dim), -exp * np.log(10))
    ret[np.arange(dim), np.arange(dim)] = 0
    return ret


_ENERGY_TRANS = log_trans_exp(150, cost0=-5)


def viterbi_decoding(emission, transition):
    """pyannote_viterbi.py:118-224 (unconstrained path) via the compiled host routine."""
    return _native.viterbi(emission, transition)


_NEP50 = int(np.__version__.split('.')[0]) >= 2      # NumPy >= 2: python-scalar / 0-d promotion by dtype, not by value


def _energy_activity(loge, ratio):
    """segmenter.py:69-73."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')                 # all-silent input: mean of an empty slice
        threshold = np.mean(loge[np.isfinite(loge)]) + np.log(ratio)
    if isinstance(loge, np.ndarray) and loge.dtype == np.float32 and isinstance(threshold, np.floating):
        # the comparison, pred2logemission and the two-state Viterbi in one compiled call: same arithmetic, no (T,2) float64
        # emission array per file.  `float32_array > float64_scalar` is a float64 comparison under NumPy >= 2 (NEP 50) and a
        # float32 one under NumPy 1.x (value-based casting demotes the scalar): the compiled call compares in float64, so for
        # 1.x the threshold is rounded to float32 first -- (double)x > (double)(float)t == x > (float)t -- and the labels are
        # whatever the installed NumPy (the one the fallback line below would use) decides
        thr = np.float64(threshold) if _NEP50 else np.float64(np.float32(threshold))
        return _native.energy_viterbi(loge, thr, _ENERGY_TRANS)
    raw_activity = (loge > threshold)
    return viterbi_decoding(pred2logemission(raw_activity), log_trans_exp(150, cost0=-5))


def _binidx2seglist(binidx):
    """Run-length encode a label sequence, segmenter.py:91-108 (vectorised)."""
    a = np.asarray(binidx)
    n = len(a)
    if n == 0:                                          # (the reference's loop dies with UnboundLocalError on [])
        raise ValueError('_binidx2seglist: empty label sequence')
    cut = np.flatnonzero(a[1:] != a[:-1]) + 1
    starts = np.concatenate(([0], cut))
    stops = np.concatenate((cut, [n]))
    return [(a[s].item() if hasattr(a[s], 'item') else a[s], int(s), int(e)) for s, e in zip(starts, stops)]


def _window_rows(nframes, difflen=0):
    """First mel row of the 68-frame window behind every 20 ms slot.

    Restates the index arithmetic of `_get_patches(mspec, 68, 2)` (segmenter.py:76-88):
    windows start every 2 frames; 17 copies of the first window are prepended and
    16 (+1 if the frame count is odd) copies of the last are appended (:83-84); for media
    shorter than 68 frames the last int(difflen/2) slots are dropped (:150-152)."""
    nwin = (nframes - 68) // 2 + 1
    nslots = nwin + 17 + 16 + (nframes % 2)
    if difflen > 0:
        nslots -= int(difflen / 2)
    w = np.clip(np.arange(nslots) - 17, 0, nwin - 1)
    return (2 * w).astype(np.int32)


# ---------------------------------------------------------------- model location
def locate_model(model_fname):
    """Same search order as remote_utils.py:18-27 (`/root/.keras/inaSpeechSegmenter/` first, then
    `~/.keras/inaSpeechSegmenter/`), plus the flat `.npz` export next to it.  There is no
    download step: the target machines have no network."""
    stem = os.path.splitext(model_fname)[0]
    for d in _MODEL_DIRS:
        for cand in (model_fname, stem + '.npz'):
            p = os.path.join(d, cand)
            if os.access(p, os.R_OK):
                return p
    raise FileNotFoundError(
        f"model file {model_fname} not found in {_MODEL_DIRS}. Download it from "
        f"https://github.com/ina-foss/inaSpeechSegmenter/releases/download/models/{model_fname} "
        "(or its tools/convert_keras_hdf5.py .npz export) into one of these directories, or "
        "construct Segmenter(..., models='synthetic') for seeded stand-in weights.")


class DnnSegmenter:
    """Mirror of segmenter.py:111-179.  Child classes define outlabels / model_fname / inlabel /
    nmel / viterbi_arg exactly as the reference does (:182-204)."""
    net_id = 0

    def __init__(self, batch_size, ctx=None, models=None):
        self.batch_size = batch_size                   # kept for API compatibility; the engine sizes
        self.ctx = ctx                                 # its own passes from the HBM workspace limit
        if models == 'synthetic':
            layers, in_shape = keras_model.synthetic_ina_like(self.nmel, len(self.outlabels), seed=self.net_id + 1)
        elif isinstance(models, dict) and self.model_fname in models:
            layers, in_shape = models[self.model_fname]
        else:
            layers, in_shape = keras_model.load_model_file(locate_model(self.model_fname))
        if tuple(in_shape) != (68, self.nmel, 1):
            raise ValueError(f"{self.model_fname}: input shape {in_shape}, expected (68, {self.nmel}, 1)")
        self.layers = layers
        self.compiled = keras_model.compile_layers(layers, in_shape, patch_input=True)
        if self.compiled.out_dim != len(self.outlabels):
            raise ValueError(f"{self.model_fname}: {self.compiled.out_dim} outputs for labels {self.outlabels}")
        if ctx is not None:
            ctx.cnn_load(self.net_id, self.compiled)

    def predict_slots(self, win_rows):
        """(n,C) float32 probabilities (+ finite mask) for the given slots of the resident mspec."""
        return self.probs(self.ctx, win_rows)

    _MODES = {'bf16x3': _native.PREC_BF16X3, 'f32': _native.PREC_F32, 'f16x3': _native.PREC_F16X3}

    def probs(self, ctx, win_rows, async_out=None):
        """iss_cnn_probs (or, with async_out = (probs, finite) page-locked arrays, iss_cnn_probs_async) of this network on `ctx`.
        The library's precision guard (include/iss.h) decides a network's arithmetic at its first call PER CONTEXT; the device
        contexts of one Segmenter (its own and the pipeline workers') must not decide differently, so the first call anywhere
        decides for all of them: it runs under a lock, and every other context is told the outcome before its first call."""
        def run():
            if async_out is None:
                return ctx.cnn_probs(self.net_id, win_rows)
            return ctx.cnn_probs_async(self.net_id, win_rows, *async_out)
        st = self.__dict__.setdefault('_mode_state', {'lock': threading.Lock(), 'mode': None, 'told': set()})
        if not hasattr(ctx, 'cnn_precision_info'):             # (a test double of the device context)
            return run()
        if st['mode'] is None:
            with st['lock']:
                if st['mode'] is None:
                    out = run()
                    info = ctx.cnn_precision_info(self.net_id)
                    # 'pending' after a call = guard off: nothing to agree on
                    st['mode'] = -1 if info['state'] == 'pending' else self._MODES[info['mode']]
                    st['told'].add(id(ctx))
                    return out
        if st['mode'] >= 0 and id(ctx) not in st['told']:
            with st['lock']:
                if id(ctx) not in st['told']:
                    if ctx.cnn_precision_info(self.net_id)['state'] == 'pending':
                        ctx.cnn_set_net_precision(self.net_id, st['mode'])
                    st['told'].add(id(ctx))
        return run()

    def __call__(self, mspec, lseg, difflen=0, dense=False, ctx=None, allpred=None):
        """mspec: the RESIDENT mel spectrogram's frame count holder (`_Resident`) or a (T,24)
        array (uploaded first).  lseg: [(label, start, stop)] in 20 ms slots.  Returns the
        refined list, like segmenter.py:135-179.
        dense=True evaluates the network on EVERY slot of the file and then keeps the rows of
        the `inlabel` segments (same result; input-independent device work, used by bench.py).
        ctx: another device context on which this network is loaded under the same id (pipeline workers).
        allpred: probabilities of EVERY slot, already computed (Segmenter.segment_slots submits both networks
        asynchronously in dense mode and smooths the first while the device evaluates the second)."""
        ctx = ctx or self.ctx
        nframes = _ensure_resident(ctx, mspec)
        rows = _window_rows(nframes, difflen)
        todo = [(start, stop) for lab, start, stop in lseg if lab == self.inlabel]
        if todo:
            idx = np.concatenate([np.arange(s, e) for s, e in todo])
            if allpred is not None:
                rawpred = allpred[idx]
            elif dense:
                allpred, _finite = self.probs(ctx, rows)
                rawpred = allpred[idx]
            else:
                rawpred, _finite = self.probs(ctx, rows[idx])  # non-finite windows already at 0.5 (:175)
        ret = []
        trans = diag_trans_exp(self.viterbi_arg, len(self.outlabels))
        pos = 0
        for lab, start, stop in lseg:
            if lab != self.inlabel:
                ret.append((lab, start, stop))
                continue
            n = stop - start
            r = rawpred[pos:pos + n]
            pos += n
            with np.errstate(divide='ignore'):
                pred = viterbi_decoding(np.log(r), trans)
            for lab2, start2, stop2 in _binidx2seglist(pred):
                ret.append((self.outlabels[int(lab2)], start2 + start, stop2 + start))
        return ret


class SpeechMusic(DnnSegmenter):
    outlabels = ('speech', 'music')
    model_fname = 'keras_speech_music_cnn.hdf5'
    inlabel = 'energy'
    nmel = 21
    viterbi_arg = 150
    net_id = 0


class SpeechMusicNoise(DnnSegmenter):
    outlabels = ('speech', 'music', 'noise')
    model_fname = 'keras_speech_music_noise_cnn.hdf5'
    inlabel = 'energy'
    nmel = 21
    viterbi_arg = 80
    net_id = 0


class Gender(DnnSegmenter):
    outlabels = ('female', 'male')
    model_fname = 'keras_male_female_cnn.hdf5'
    inlabel = 'speech'
    nmel = 24
    viterbi_arg = 80
    net_id = 1


class _Resident:
    """Marker for "the mel spectrogram is already in HBM on this context"."""
    def __init__(self, ctx, nframes):
        self.ctx, self.nframes = ctx, nframes

    def __len__(self):
        return self.nframes

    def to_host(self):
        return self.ctx.get_mspec()


def _ensure_resident(ctx, mspec):
    if isinstance(mspec, _Resident):
        if mspec.ctx is not ctx:
            raise ValueError("mel spectrogram is resident on a different context")
        return mspec.nframes
    m = np.asarray(mspec)
    ctx.set_mspec(m.astype(np.float32))
    return m.shape[0]


def _media2feats(medianame, start_sec, stop_sec, ffmpeg, ctx=None):
    """segmenter.py:53-67 on the device: returns (mspec, loge, difflen) where mspec is a
    `_Resident` handle when a context is given (the (T,24) array stays in HBM)."""
    if ctx is None:
        raise _native.NativeError("feature extraction needs a device context (no CPU path)")
    sig = decode_pcm(medianame, start_sec, stop_sec, ffmpeg)
    return _sig2feats(ctx, sig, medianame)


def _sig2feats(ctx, sig, medianame='<signal>'):
    if sig.size < 400:
        raise ValueError(f"media {medianame}: {sig.size} samples, less than one 25 ms analysis window")
    ctx.set_signal(sig)
    nframes = ctx.sidekit()
    loge = ctx.get_loge()
    difflen = 0
    if nframes < 68:                                            # segmenter.py:61-65
        difflen = 68 - nframes
        warnings.warn("media %s duration is short. Robust results require length of at least 720 milliseconds" % medianame)
        mspec = ctx.get_mspec()
        mspec = np.concatenate((mspec, np.ones((difflen, 24)) * np.min(mspec)))
        ctx.set_mspec(mspec.astype(np.float32))
        nframes = 68
    return _Resident(ctx, nframes), loge, difflen


class Segmenter:
    def __init__(self, vad_engine='smn', detect_gender=True, ffmpeg='ffmpeg', batch_size=32, energy_ratio=0.03,
                 device=0, models=None):
        """Load the networks onto one MI355X.

        vad_engine / detect_gender / ffmpeg / batch_size / energy_ratio: as segmenter.py:208-247
        (same assertions, same "ffmpeg program not found" exception).
        device: HIP device ordinal.  models: None -> Keras files from ~/.keras/inaSpeechSegmenter
        (remote_utils.py search path); 'synthetic' -> seeded stand-in weights; or a dict
        {model_fname: (layers, in_shape)}."""
        if ffmpeg is not None:
            if shutil.which(ffmpeg) is None:
                raise (Exception("""ffmpeg program not found"""))
        self.ffmpeg = ffmpeg
        self.energy_ratio = energy_ratio
        self.dense_batches = False                     # extension: batch_process evaluates both networks on every slot (pipeline.py)

        self.ctx = _native.Context(device)
        self.ctx.sidekit_tables(tables.sidekit_window(), tables.sidekit_melbank())

        assert vad_engine in ['sm', 'smn']
        if vad_engine == 'sm':
            self.vad = SpeechMusic(batch_size, self.ctx, models)
        elif vad_engine == 'smn':
            self.vad = SpeechMusicNoise(batch_size, self.ctx, models)

        assert detect_gender in [True, False]
        self.detect_gender = detect_gender
        if detect_gender:
            self.gender = Gender(batch_size, self.ctx, models)

    def close(self):
        """Release the extra device contexts of the multi-file pipeline (the main context goes with the object)."""
        from . import pipeline
        pipeline.close_workers(self)

    def segment_slots(self, mspec, loge, difflen, dense=False):
        """The body of segmenter.py:250-275 in 20 ms slot units: [(label, start_slot, stop_slot)]."""
        pending = None
        if dense and self.detect_gender:
            # both networks on every slot, independent of each other and of the energy detector: enqueue them back to back
            # (iss_cnn_probs_async) BEFORE the host starts on the energy Viterbi, and smooth the VAD output while the device
            # evaluates the gender network (the reference overlaps host and device work across files, segmenter.py:377-387;
            # within one file its gender pass needs the VAD result)
            ctx = self.ctx
            rows = _window_rows(_ensure_resident(ctx, mspec), difflen)
            t1, p1, _ = self.vad.probs(ctx, rows, async_out=self._pinned_out(0, len(rows), len(self.vad.outlabels)))
            t2, p2, _ = self.gender.probs(ctx, rows, async_out=self._pinned_out(1, len(rows), len(self.gender.outlabels)))
            pending = (t1, p1, t2, p2)
        lseg = []
        for lab, start, stop in _binidx2seglist(_energy_activity(loge, self.energy_ratio)[::2]):
            lseg.append(('noEnergy' if lab == 0 else 'energy', start, stop))
        if pending is not None:
            t1, p1, t2, p2 = pending
            self.ctx.wait(t1)
            lseg = self.vad(mspec, lseg, difflen, allpred=p1)
            self.ctx.wait(t2)
            return self.gender(mspec, lseg, difflen, allpred=p2)
        lseg = self.vad(mspec, lseg, difflen, dense=dense)
        if self.detect_gender:
            lseg = self.gender(mspec, lseg, difflen, dense=dense)
        return lseg

    def _pinned_out(self, k, n, c):
        """Page-locked result arrays of the asynchronous dense path (grown on demand, one set per network)."""
        cache = self.__dict__.setdefault('_pin_out', {})
        cur = cache.get(k)
        if cur is None or cur[0].shape[0] < n or cur[0].shape[1] != c:
            if cur is not None:
                self.ctx.pinned_free(cur[0]); self.ctx.pinned_free(cur[1])
            cap = int(n * 1.1) + 64
            cur = (self.ctx.pinned_empty((cap, c), np.float32), self.ctx.pinned_empty((cap,), np.uint8))
            cache[k] = cur
        return cur[0][:n], cur[1][:n]

    def segment_feats(self, mspec, loge, difflen, start_sec):
        """segmenter.py:250-276.  `mspec` may be a (T,24) array or the resident handle."""
        lseg = self.segment_slots(mspec, loge, difflen)
        return [(lab, start_sec + start * .02, start_sec + stop * .02) for lab, start, stop in lseg]

    def segment_device_pcm(self, dev_ptr, n, dense=False):
        """Hot-path entry for PCM16 samples that are ALREADY in this GPU's HBM (`dev_ptr` = a
        hipMalloc'ed address, e.g. a torch int16 tensor's data_ptr(); it must stay alive during
        the call).  Returns slot-unit segments [(label, start_slot, stop_slot)]."""
        if n < 400 + 160 * 67:
            raise ValueError("segment_device_pcm needs at least 68 frames (use segment_signal for short media)")
        self.ctx.set_signal_device(dev_ptr, n)
        nframes = self.ctx.sidekit()
        loge = self.ctx.get_loge()
        return self.segment_slots(_Resident(self.ctx, nframes), loge, 0, dense=dense)

    def segment_signal(self, sig, start_sec=0):
        """Native-path entry for an already decoded 16 kHz mono signal (int16 or float32)."""
        mspec, loge, difflen = _sig2feats(self.ctx, np.ascontiguousarray(sig))
        return self.segment_feats(mspec, loge, difflen, start_sec)

    def __call__(self, medianame, start_sec=None, stop_sec=None):
        """segmenter.py:279-294."""
        mspec, loge, difflen = _media2feats(medianame, start_sec, stop_sec, self.ffmpeg, self.ctx)
        if start_sec is None:
            start_sec = 0
        return self.segment_feats(mspec, loge, difflen, start_sec)

    def batch_process(self, linput, loutput, verbose=False, skipifexist=False, nbtry=1, trydelay=2., output_format='csv',
                      batch_files=None, workers=None, batch_seconds=None):
        """segmenter.py:297-335: same arguments, same (t_batch_dur, nb_processed, avg, lmsg) with
        lmsg entries (dst, code, text), code 0 ok / 1 already exists / 2 error, in input order.
        The files run through pipeline.process_files: decode threads, super-batches of at most `batch_files` files /
        `batch_seconds` of audio per device pass (defaults 32 / 40 min), `workers` (default 4) device contexts in turn (the reference overlaps the feature extraction of file i+1 with the
        networks of file i, :377-387).  Undecodable or too-short media are per-file errors (code 2) as in the
        reference; device failures and unwritable outputs raise."""
        from . import pipeline
        if verbose:
            print('batch_processing %d files' % len(linput))
        if output_format == 'csv':
            fexport = seg2csv
        elif output_format == 'textgrid':
            fexport = seg2textgrid
        else:
            raise NotImplementedError()

        t_batch_start = time.time()
        linput, loutput = list(linput), list(loutput)
        msgs, skip, done = {}, set(), [0]
        for i, dst in enumerate(loutput):
            if skipifexist and os.path.exists(dst):
                msgs[i] = (dst, 1, 'already exists')
                skip.add(i)
            else:
                dname = os.path.dirname(dst)
                if dname and not os.path.isdir(dname):
                    os.makedirs(dname, exist_ok=True)

        def on_result(i, src, lseg, err, secs=0.0):
            dst = loutput[i]
            if lseg is None:
                msgs[i] = (dst, 2, err)
            else:
                b = time.time()
                fexport(lseg, dst)
                # per-file processing time as in segmenter.py:322-327 (networks + smoothing + export, features excluded
                # there; here: this file's share of its device pass + its export)
                msgs[i] = (dst, 0, 'ok ' + str(secs + time.time() - b))
            done[0] += 1
            if verbose:
                print('%d/%d' % (done[0] + len(skip), len(linput)), [msgs[i]])

        pipeline.process_files(self, linput, on_result, skip=skip, nbtry=nbtry, trydelay=trydelay,
                               batch_files=batch_files, workers=workers, batch_seconds=batch_seconds)
        lmsg = [msgs[i] for i in range(len(linput))]
        t_batch_dur = time.time() - t_batch_start
        nb_processed = len([e for e in lmsg if e[1] == 0])
        avg = t_batch_dur / nb_processed if nb_processed > 0 else -1
        return t_batch_dur, nb_processed, avg, lmsg
