"""Constant tables uploaded once per device context (host side, numpy).

They are built with the same numpy expressions the reference evaluates so that the values
(and, for the mel banks, the floor() bin decisions) are the ones the reference gets on the
same host: sidekit_mfcc.py:118-197 (trfbank, nlinfilt=0 branch), :223 (numpy.hanning);
features_vbx.py:31-59 (mel_fbank_mx, htk_bug=False), :123-124 (povey_window).
tests/test_host.py::test_tables_match_reference compares them with the committed reference outputs.
"""
import numpy as np


def sidekit_window():
    return np.hanning(400)


def sidekit_melbank(fs=16000, nfft=512, lowfreq=100.0, maxfreq=8000.0, nfilt=24):
    """(24, 257) float32.  HTK mel 2595*log10(1+f/700); nfilt+2 edges equally spaced in mel;
    triangle i rises over bins floor(lo*nfft/fs)+1 .. floor(cen*nfft/fs) and falls over
    floor(cen*nfft/fs)+1 .. floor(hi*nfft/fs)-1 (the last falling bin is dropped, :195);
    peak height 2/(hi-lo)."""
    to_mel = lambda f: 2595 * np.log10(1 + f / 700.)
    mels = np.zeros(nfilt + 2)
    mels[:] = to_mel(lowfreq) + np.arange(nfilt + 2) * ((to_mel(maxfreq) - to_mel(lowfreq)) / (nfilt + 1))
    hz = 700. * (10 ** (mels / 2595.) - 1)
    peak = 2. / (hz[2:] - hz[:-2])
    bin_hz = np.arange(nfft) / (1. * nfft) * fs
    bank = np.zeros((nfilt, nfft // 2 + 1), dtype=np.float32)
    for f in range(nfilt):
        lo, cen, hi = hz[f], hz[f + 1], hz[f + 2]
        b_lo = int(np.floor(lo * nfft / fs)) + 1
        b_cen = int(np.floor(cen * nfft / fs)) + 1
        b_hi = int(min(np.floor(hi * nfft / fs) + 1, nfft))
        up = np.arange(b_lo, b_cen, dtype=np.int32)
        down = np.arange(b_cen, b_hi, dtype=np.int32)[:-1]
        bank[f][up] = peak[f] / (cen - lo) * (bin_hz[up] - lo)
        bank[f][down] = peak[f] / (hi - cen) * (hi - bin_hz[down])
    return bank


def vbx_window(n=400):
    return np.power(0.5 - 0.5 * np.cos(np.linspace(0, 2 * np.pi, n)), 0.85)


def vbx_melbank(nfft=512, fs=16000, nch=64, lofreq=20.0, hifreq=7600.0):
    """(257, 64) float64, mel = 1127*ln(1+f/700)."""
    warp = lambda x: 1127. * np.log(1. + x / 700.)
    unwarp = lambda x: (np.exp(x / 1127.) - 1.) * 700.
    fbin = warp(np.arange(nfft / 2 + 1, dtype=float) * fs / nfft)
    cbin = np.linspace(warp(lofreq), warp(hifreq), nch + 2)
    cind = np.floor(unwarp(cbin) / fs * nfft).astype(int) + 1
    bank = np.zeros((len(fbin), nch))
    for i in range(nch):
        a, b, c = cind[i], cind[i + 1], cind[i + 2]
        bank[a:b, i] = (cbin[i] - fbin[a:b]) / (cbin[i] - cbin[i + 1])
        bank[b:c, i] = (cbin[i + 2] - fbin[b:c]) / (cbin[i + 2] - cbin[i + 1])
    return bank
