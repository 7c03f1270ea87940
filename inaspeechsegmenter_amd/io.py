"""Media decode boundary: anything -> 16 kHz mono samples on the host.

Mirrors io.py:32-79 `media2sig16kmono(medianame, start_sec, stop_sec, ffmpeg, dtype)`:
  * ffmpeg given  -> `ffmpeg -i <media> -f wav -acodec pcm_s16le -ar 16000 -ac 1 [-ss] [-to] pipe:1`
    (same command line, io.py:61-68); non-zero exit raises Exception(stderr) (io.py:72-75).
    The PCM is taken straight from the pipe instead of a TemporaryFile + soundfile.
  * ffmpeg=None   -> direct read; start/stop and http(s) sources raise NotImplementedError
    with the reference's wording (io.py:37-50); the file must already be 16 kHz (io.py:53-55).
Decode stays on the host (north star); soundfile/libsndfile is not available in the target
image, so RIFF/WAVE is parsed here with libsndfile's conversion rules (PCM16 -> x/32768,
unknown chunks skipped).  `decode_pcm` is the entry the native pipeline uses: it keeps
PCM16 as int16 so the device does the x/32768 scaling (2 B/sample over PCIe, not 4).
"""
import os
import struct
import subprocess

import numpy as np


def _parse_wav(buf, name='<buffer>'):
    """-> (samples ndarray shaped (n,) or (n,ch), sr).  dtype int16 / uint8 / int32 / float32 / float64
    exactly as stored (24-bit widened to int32 << 8)."""
    if len(buf) < 12 or buf[:4] not in (b'RIFF', b'RF64') or buf[8:12] != b'WAVE':
        raise ValueError(f'{name}: not a RIFF/WAVE file')
    pos, fmt, data = 12, None, None
    n = len(buf)
    while pos + 8 <= n:
        cid, size = buf[pos:pos + 4], struct.unpack_from('<I', buf, pos + 4)[0]
        body = pos + 8
        if cid == b'fmt ':
            tag, ch, sr, _, _, bits = struct.unpack_from('<HHIIHH', buf, body)
            if tag == 0xFFFE and size >= 26:                      # WAVE_FORMAT_EXTENSIBLE
                tag = struct.unpack_from('<H', buf, body + 24)[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b'data':
            end = n if (size == 0xFFFFFFFF or body + size > n) else body + size   # piped WAVs carry no length
            data = buf[body:end]
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f'{name}: missing fmt or data chunk')
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 16:
            a = np.frombuffer(data, dtype='<i2', count=len(data) // 2)
        elif bits == 8:
            a = np.frombuffer(data, dtype=np.uint8)
        elif bits == 32:
            a = np.frombuffer(data, dtype='<i4', count=len(data) // 4)
        elif bits == 24:
            raw = np.frombuffer(data, dtype=np.uint8, count=(len(data) // 3) * 3).reshape(-1, 3)
            a = ((raw[:, 0].astype(np.int32) << 8) | (raw[:, 1].astype(np.int32) << 16) | (raw[:, 2].astype(np.int32) << 24))
        else:
            raise ValueError(f'{name}: unsupported PCM width {bits}')
    elif tag == 3:
        a = np.frombuffer(data, dtype='<f4' if bits == 32 else '<f8', count=len(data) // (bits // 8))
    else:
        raise ValueError(f'{name}: unsupported WAVE format tag {tag}')
    if ch > 1:
        a = a[:(len(a) // ch) * ch].reshape(-1, ch)
    return a, sr


def _to_float(a, dtype):
    """libsndfile's integer -> float scaling."""
    if a.dtype == np.int16:
        return (a.astype(np.float64) / 32768.0).astype(dtype)
    if a.dtype == np.uint8:
        return ((a.astype(np.float64) - 128.0) / 128.0).astype(dtype)
    if a.dtype == np.int32:
        return (a.astype(np.float64) / 2147483648.0).astype(dtype)
    return a.astype(dtype)


def _run_ffmpeg(medianame, start_sec, stop_sec, ffmpeg):
    cmd = [ffmpeg, '-i', medianame, '-f', 'wav', '-acodec', 'pcm_s16le', '-ar', '16000', '-ac', '1']
    if start_sec is not None:
        cmd += ['-ss', '%f' % start_sec]
    if stop_sec is not None:
        cmd += ['-to', '%f' % stop_sec]
    cmd += ['pipe:1']
    ret = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if ret.returncode != 0:
        raise Exception(ret.stderr)
    a, fs = _parse_wav(ret.stdout, medianame)
    assert fs == 16000
    return a


def _check_no_ffmpeg(medianame, start_sec, stop_sec):
    if start_sec is not None or stop_sec is not None:
        raise NotImplementedError(
            f'start_sec={start_sec} and stop_sec={stop_sec} cannot be set '
            f' when running inaSpeechSegmenter without ffmpeg. Please cut '
            f'down your audio files beforehand or use ffmpeg.')
    if medianame.startswith('http://') or medianame.startswith('https://'):
        raise NotImplementedError(
            f'Without ffmpeg you cannot process media content on http '
            f'servers. You need to download your audio files beforehand '
            f'or use ffmpeg. You gave medianame={medianame}.')


def decode_pcm(medianame, start_sec=None, stop_sec=None, ffmpeg='ffmpeg'):
    """16 kHz mono samples for the device: int16 when the source is PCM16 (always, through
    ffmpeg), else float32 holding exactly what soundfile's float32 read would return."""
    if ffmpeg is None:
        _check_no_ffmpeg(medianame, start_sec, stop_sec)
        with open(medianame, 'rb') as f:
            a, sr = _parse_wav(f.read(), medianame)
        assert sr == 16_000, \
            f'Without ffmpeg, inaSpeechSegmenter can only take files sampled ' \
            f'at 16000 Hz. The file {medianame} is sampled at {sr} Hz.'
        if a.ndim != 1:
            raise ValueError(f'{medianame}: {a.shape[1]} channels; without ffmpeg only mono files are supported')
    else:
        a = _run_ffmpeg(medianame, start_sec, stop_sec, ffmpeg)
    if a.dtype == np.int16:
        return np.ascontiguousarray(a)
    return np.ascontiguousarray(_to_float(a, np.float32))


def media2sig16kmono(medianame, start_sec=None, stop_sec=None, ffmpeg='ffmpeg', dtype='float64'):
    """Reference-compatible signature and result (float array of `dtype`)."""
    a = decode_pcm(medianame, start_sec, stop_sec, ffmpeg)
    return _to_float(a, np.dtype(dtype)) if a.dtype == np.int16 else a.astype(dtype)
