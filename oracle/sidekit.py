"""Oracle: SIDEKIT log-mel front end (numpy restatement).  Test infrastructure only.

Follows /root/reference/inaSpeechSegmenter/sidekit_mfcc.py:
  trfbank 118-197 (only the nlinfilt == 0 branch is exercised by mfcc's
  defaults), power_spectrum 200-237, framing 240-263, pre_emphasis 266-275,
  mfcc 278-352 (get_mspec=True path; the DCT at :337 is discarded by every
  caller in segmenter.py and is not restated).

dtype plan of the reference (and of this file), which decides labels:
  frames, pre-emphasis, energy, log-energy ........ float32
  Hann window * frame, rfft-512, re^2+im^2 ........ float64, stored as float32
  spec @ fbank.T, log ............................. float32
"""
import numpy as np

FS = 16000
WIN = 400          # int(round(0.025 * 16000))            sidekit_mfcc.py:214
HOP = 160          # int(0.01 * 16000)                    sidekit_mfcc.py:215
NFFT = 512         # 2 ** ceil(log2(400))                 sidekit_mfcc.py:220
NBIN = NFFT // 2 + 1
NMEL = 24
PREFAC = 0.97
CHUNK = 500000     # frames per rfft call                 sidekit_mfcc.py:227


def hz2mel_htk(f):
    # sidekit_mfcc.py:61-62
    return 2595 * np.log10(1 + f / 700.)


def mel2hz_htk(z):
    # sidekit_mfcc.py:93-94
    return 700. * (10 ** (z / 2595.) - 1)


def mel_bank(fs=FS, nfft=NFFT, lowfreq=100, maxfreq=8000, nfilt=NMEL):
    """(nfilt, nfft//2+1) float32 triangular bank, sidekit_mfcc.py:118-197 with
    nlinfilt=0 (branch :143-151)."""
    mels = np.zeros(nfilt + 2)
    lo, hi = hz2mel_htk(lowfreq), hz2mel_htk(maxfreq)
    step = (hi - lo) / (nfilt + 1)
    mels[:nfilt + 2] = lo + np.arange(nfilt + 2) * step
    edges = mel2hz_htk(mels)                                  # float64 here (:151)
    heights = 2. / (edges[2:] - edges[0:-2])                  # :177
    bank = np.zeros((nfilt, nfft // 2 + 1), dtype=np.float32)  # :180
    bin_hz = np.arange(nfft) / (1. * nfft) * fs                # :182
    for i in range(nfilt):
        low, cen, top = edges[i], edges[i + 1], edges[i + 2]
        lid = np.arange(np.floor(low * nfft / fs) + 1,
                        np.floor(cen * nfft / fs) + 1, dtype=np.int32)          # :189
        rid = np.arange(np.floor(cen * nfft / fs) + 1,
                        min(np.floor(top * nfft / fs) + 1, nfft), dtype=np.int32)  # :191-192
        bank[i][lid] = heights[i] / (cen - low) * (bin_hz[lid] - low)            # :190,194
        bank[i][rid[:-1]] = heights[i] / (top - cen) * (top - bin_hz[rid[:-1]])  # :193,195
    return bank, edges


def num_frames(n):
    # framing(): int((n - win)/shift) + 1, sidekit_mfcc.py:254
    return int((n - WIN) / HOP) + 1 if n >= WIN else 0


def frames_of(sig):
    """(T, 400) copy of the hop-160 frames (sidekit_mfcc.py:216,240-263; context (0,0))."""
    t = num_frames(len(sig))
    idx = np.arange(WIN)[None, :] + HOP * np.arange(t)[:, None]
    return sig[idx]


def pre_emphasis_per_frame(fr, pre=PREFAC):
    # sidekit_mfcc.py:275 -- frame-local: y[0] = x[0] - pre*x[0]
    shifted = np.concatenate([fr[:, :1], fr[:, :-1]], axis=1)
    return fr - shifted * pre


def power_spectrum(sig):
    """-> (spec (T,257) f32, log_energy (T,) f32), sidekit_mfcc.py:200-237."""
    fr = pre_emphasis_per_frame(frames_of(sig))
    t = fr.shape[0]
    window = np.hanning(WIN)                                   # float64, :223
    spec = np.ones((t, NBIN), dtype=np.float32)                # :225
    with np.errstate(divide='ignore'):
        log_energy = np.log((fr ** 2).sum(axis=1))             # :226 (float32)
    for s in range(0, t, CHUNK):                               # :227-235
        e = min(s + CHUNK, t)
        mag = np.fft.rfft(fr[s:e, :] * window, NFFT, axis=-1)
        spec[s:e, :] = mag.real ** 2 + mag.imag ** 2
    return spec, log_energy


_BANK = None


def mfcc_mspec(sig):
    """Reference mfcc(sig, get_mspec=True) restricted to what the callers keep:
    returns (loge (T,) f32, mspec (T,24) f32).  sidekit_mfcc.py:325-334."""
    global _BANK
    sig = np.asarray(sig)
    assert sig.dtype == np.float32, "segmenter.py:58 casts to float32 first"
    if _BANK is None:
        _BANK = mel_bank()[0]
    spec, loge = power_spectrum(sig)
    with np.errstate(divide='ignore'):
        mspec = np.log(np.dot(spec, _BANK.T))                  # :334
    return loge, mspec


def media2feats(sig):
    """segmenter.py:53-67 minus the decode: -> (mspec, loge, difflen)."""
    loge, mspec = mfcc_mspec(np.asarray(sig).astype(np.float32))
    difflen = 0
    if len(loge) < 68:                                         # :62-65
        difflen = 68 - len(loge)
        mspec = np.concatenate((mspec, np.ones((difflen, 24)) * np.min(mspec)))
    return mspec, loge, difflen
