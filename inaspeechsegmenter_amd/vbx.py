"""Host mirror of the x-vector half of vbx_segmenter.py on the gfx950 kernels (config 5).

Mirrors, for the hot path only:
  get_features(signal)                 vbx_segmenter.py:72-89   -> iss_vbx_features (csrc/vbx.hip)
  VBxExtractor.__call__(basename, fea, duration)   :217-246     -> window bookkeeping here,
  OnnxBackendExtractor.get_embedding(fea)          :262-266        ResNet-101 of resnet.py:78-135 as an
                                                                   op program run by iss_cnn_forward, many
                                                                   windows per launch sequence instead of
                                                                   one onnxruntime call per window.
The VAD filtering / MLP scoring tail of VoiceFemininityScoring (:129-202) lives in vfs.py.
"""
import logging
import os

import numpy as np

from . import _native
from . import tables
from . import keras_model

STEP = 24        # vbx_segmenter.py:21
WINLEN = 144     # vbx_segmenter.py:22
FEAT_DIM = 64    # vbx_segmenter.py:23
EMBED_DIM = 256  # vbx_segmenter.py:24
SR = 16000       # vbx_segmenter.py:25

logger = logging.getLogger(__name__)


def dither_stream(n, seed=3):
    """`np.random.seed(3)` + `np.random.rand(n)` (vbx_segmenter.py:84, features_vbx.py:127-128):
    an MT19937 stream, generated on the host and uploaded."""
    return np.random.RandomState(seed).rand(n)


class ResidentFeatures:
    """Stand-in for the (T, 64) feature array when it is left on the device (`FeatureExtractor(..., to_host=False)`):
    VBxExtractor only needs its length and identity, the windows are gathered on the device (iss_vbx_embed)."""

    def __init__(self, nframes):
        self.nframes = nframes

    def __len__(self):
        return self.nframes


class FeatureExtractor:
    """get_features (vbx_segmenter.py:72-89) bound to one device context.  The features also stay resident on
    the device; `VBxExtractor` recognises the array it is handed back and gathers its windows there."""

    def __init__(self, ctx):
        self.ctx = ctx
        ctx.vbx_tables(tables.vbx_window(), tables.vbx_melbank())

    def _ensure_dither(self, n):
        """np.random.seed(3) restarts the stream for every file (vbx_segmenter.py:84): all files share one stream,
        so its longest prefix lives on the device."""
        if n > self.ctx._dither_n:
            self.ctx.vbx_set_dither(dither_stream(max(n, 2 * self.ctx._dither_n)))

    def __call__(self, signal, to_host=True):
        """to_host=False (PCM16 input only): the (T, 64) array is not copied back (92 MB per audio-hour), a
        `ResidentFeatures` handle is returned instead."""
        signal = np.asarray(signal)
        if signal.dtype == np.int16:                                  # PCM16 source: (signal/32768 * 2**15).astype(int) == pcm
            pcm = signal
        else:
            sig_i = (np.asarray(signal, dtype=np.float64) * 2 ** 15).astype(int)      # :85 truncation toward zero
            pcm = sig_i.astype(np.int16) if sig_i.size and -32768 <= sig_i.min() and sig_i.max() <= 32767 else None
            if pcm is None:                                           # out-of-range float input: the general int32 path
                fea = self.ctx.vbx_features(sig_i.astype(np.int32), dither_stream(len(sig_i)))
                self.ctx._vbx_resident = fea
                return fea
        self._ensure_dither(len(pcm))
        if not to_host:
            fea = ResidentFeatures(self.ctx.vbx_features_pcm16(pcm, to_host=False))
            self.ctx._vbx_resident = fea
            return fea
        fea = self.ctx.vbx_features_pcm16(pcm)
        self.ctx._vbx_resident = fea                                  # identity token for VBxExtractor
        return fea


class VBxExtractor:
    """x-vector extraction with the reference's window bookkeeping (vbx_segmenter.py:217-246).

    params: state_dict-like mapping of resnet.py's ResNet101 (conv OIHW, BN weight / bias /
    running_mean / running_var, embedding.weight / bias) as numpy arrays."""
    _NET_BASE = 4            # net ids 4.. are used for the per-length programs

    def __init__(self, ctx, params, batch_windows=256):
        self.ctx = ctx
        self.params = params
        self.batch_windows = batch_windows
        self._nets = {}      # (frames, device-window input) -> net_id
        self._lru = []       # keys of the two tail-length slots, oldest first

    def _net_for(self, frames, window=False):
        """Engine program for this window length.  The full-length programs keep fixed slots (4: host input, 5: device
        windows); the shorter last-window programs share slots 6-7, least recently used one reloaded on demand."""
        key = (frames, window)
        if key in self._nets:
            if frames != WINLEN:
                self._lru.remove(key)
                self._lru.append(key)
            return self._nets[key]
        if frames == WINLEN:
            nid = self._NET_BASE + (1 if window else 0)
        elif len(self._lru) < 2:
            nid = self._NET_BASE + 2 + len(self._lru)
        else:
            old = self._lru.pop(0)
            nid = self._nets.pop(old)
        if frames != WINLEN:
            self._lru.append(key)
        self.ctx.cnn_load(nid, keras_model.compile_resnet101(self.params, FEAT_DIM, frames, window_input=window))
        self._nets[key] = nid
        return nid

    def get_embeddings(self, fea, starts, frames):
        """(len(starts), 256) embeddings of fea[s:s+frames] (feature-major input like :265).  When `fea` is the
        array FeatureExtractor just produced on this context, the windows are gathered on the device."""
        if getattr(self.ctx, '_vbx_resident', None) is fea:
            return self.ctx.vbx_embed(self._net_for(frames, window=True), starts)
        nid = self._net_for(frames)
        out = np.empty((len(starts), EMBED_DIM), dtype=np.float32)
        for i in range(0, len(starts), self.batch_windows):
            st = starts[i:i + self.batch_windows]
            x = np.stack([fea[s:s + frames].T for s in st])[..., None]       # (n, 64, frames, 1) NHWC
            out[i:i + len(st)] = self.ctx.cnn_forward(nid, x.astype(np.float32))
        return out

    def get_embedding(self, fea):
        """Single-window form of OnnxBackendExtractor.get_embedding (vbx_segmenter.py:262-266)."""
        return self.get_embeddings(np.asarray(fea), [0], len(fea))[0]

    def __call__(self, basename, fea, duration):
        if not isinstance(fea, ResidentFeatures):
            fea = np.asarray(fea)
        xvectors = []
        starts = list(range(0, len(fea) - WINLEN, STEP))
        start = starts[-1] if starts else 0
        emb = self.get_embeddings(fea, starts, WINLEN) if starts else np.zeros((0, EMBED_DIM), np.float32)
        # the NaN test and the x 10 of :246 on the whole (windows, 256) array at once: per window they were 2 / 3 of this function's
        # 29 ms per audio-hour, during which the device idles (profiles/r06_vbx_gaps.txt); keys, rounding and order as the reference's
        bad = np.isnan(emb).any(axis=1)
        emb10 = emb * 10
        for i, s in enumerate(starts):
            key = f'{basename}_{s:08}-{(s + WINLEN):08}'
            if bad[i]:
                logger.warning(f'NaN found, not processing: {key}{os.linesep}')
            else:
                xvectors.append((key, (round(s / 100.0, 3), round(s / 100.0 + WINLEN / 100.0, 3)), emb10[i]))
        if len(fea) - start - STEP >= 10:                               # last, shorter window (:234-243)
            xvector = self.get_embeddings(fea, [start + STEP], len(fea) - start - STEP)[0]
            key = f'{basename}_{(start + STEP):08}-{len(fea):08}'
            if np.isnan(xvector).any():
                logger.warning(f'NaN found, not processing: {key}{os.linesep}')
            else:
                xvectors.append((key, (round((start + STEP) / 100.0, 3), round(duration, 3)), xvector * 10))
        return xvectors
