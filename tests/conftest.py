import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')


@pytest.fixture(scope='session')
def golden():
    return GOLDEN


def read_wav_int16(path):
    from inaspeechsegmenter_amd.io import decode_pcm
    return decode_pcm(path, ffmpeg=None)


@pytest.fixture(scope='session')
def ctx():
    """One device context shared by the GPU tests (fails loudly without a GPU)."""
    from inaspeechsegmenter_amd import _native, tables
    c = _native.Context(0)
    c.sidekit_tables(tables.sidekit_window(), tables.sidekit_melbank())
    c.vbx_tables(tables.vbx_window(), tables.vbx_melbank())
    # the precision guard (include/iss.h) probes a patch network's first call in both arithmetic modes: off on this shared
    # context, whose tests count launches and kernel names per call; test_precision_guard_* switches it on, and every
    # Segmenter the other tests construct runs with the library default (on)
    c.set_precision_guard(0)
    # ... and in the split-bf16 arithmetic every GEMM kernel has (the tests that switch kernel families with iss_set_diag expect
    # bit-identical results; the library default, fp16 halves, exists in the kernels of the segmenter nets only:
    # test_f16x3_mode / test_precision_guard_* and every Segmenter of the other tests run it)
    c.set_precision({'f16x3': _native.PREC_F16X3, 'f32': _native.PREC_F32}.get(os.environ.get('ISS_TEST_PREC', ''), _native.PREC_BF16X3))
    yield c
    c.close()


def synth_pcm(seed, n):
    """Same generator as tests/golden/make_golden.py::synth_signal."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = np.zeros(n)
    q = n // 4
    x[q:2 * q] = rng.normal(0, 0.03, q)
    x[2 * q:3 * q] = sum(0.1 / k * np.sin(2 * np.pi * 110 * k * t[2 * q:3 * q]) for k in range(1, 20)) \
        * (0.6 + 0.4 * np.sin(2 * np.pi * 4 * t[2 * q:3 * q]))
    x[3 * q:] = 0.05 * (np.sin(2 * np.pi * 440 * t[3 * q:]) + np.sin(2 * np.pi * 554.37 * t[3 * q:])) \
        + rng.normal(0, 0.001, n - 3 * q)
    return np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
