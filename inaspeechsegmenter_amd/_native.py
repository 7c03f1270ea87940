"""ctypes binding of libiss_hip.so (include/iss.h).  No fallback: if the shared library is
missing, or a device call is made without a usable gfx950 GPU, this raises.

The library is built in-tree by `__graft_entry__.build()` (hipcc, --offload-arch=gfx950).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ISS_LIB') or os.path.join(_HERE, 'libiss_hip.so')      # ISS_LIB: another build of the library (A/B runs)

PROG_COLS = 32
OP_CONV, OP_POOL, OP_SOFTMAX, OP_STATPOOL, OP_ACT, OP_ELT = 1, 2, 3, 4, 5, 6
(ELT_COPY, ELT_ADD, ELT_SUB, ELT_MUL, ELT_MAX, ELT_MIN, ELT_AVG, ELT_ZERO, ELT_PERMUTE) = range(9)     # ISS_OP_ELT kinds (ISS_C_ACT)
(C_OP, C_IN, C_OUT, C_RES, C_H, C_W, C_CIN, C_HO, C_WO, C_COUT, C_KH, C_KW, C_SH, C_SW, C_PT, C_PL,
 C_ACT, C_WOFF, C_BOFF, C_PSOFF, C_PTOFF, C_INMODE, C_POOLKIND, C_ORDER, C_FPOOLH, C_FPOOLW, C_DUALW, C_DUALB, C_ACTPARAM, C_ACTPARAM2, C_ACTPARAM3) = range(31)
PREC_BF16X3, PREC_F32, PREC_F16X3 = 0, 1, 2
K_ALIGN = 32          # conv weight rows are padded to a multiple of this many k
BUF_INPUT = -2
MAX_NETS = 8


class NativeError(RuntimeError):
    pass


_lib = None


def lib():
    """The loaded shared library; raises NativeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback for the feature/CNN path.")
    # PyTorch-ROCm ships its own HIP runtime (libamdhip64 with the same SONAME as /opt/rocm's).  Two HIP runtimes
    # in one process do not both see the GPU ("No HIP GPUs are available" in whichever initialises second), so make
    # the order deterministic: if torch is installed, let it load its runtime first and bind to that one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    pf, pd, pi32, pi64, pu8, pi16 = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_int16))
    sig = {
        'iss_create': (C.c_int, [C.c_int, C.POINTER(vp)]),
        'iss_destroy': (None, [vp]),
        'iss_last_error': (C.c_char_p, [vp]),
        'iss_version': (C.c_char_p, []),
        'iss_set_workspace_limit': (C.c_int, [vp, u64]),
        'iss_synchronize': (C.c_int, [vp]),
        'iss_sidekit_tables': (C.c_int, [vp, pd, pf]),
        'iss_signal_pcm16': (C.c_int, [vp, pi16, i64]),
        'iss_signal_f32': (C.c_int, [vp, pf, i64]),
        'iss_signal_pcm16_device': (C.c_int, [vp, vp, i64]),
        'iss_signal_pcm16_device_stream': (C.c_int, [vp, vp, i64, vp]),
        'iss_host_alloc': (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        'iss_host_free': (C.c_int, [vp, vp]),
        'iss_cnn_probs_async': (C.c_int, [vp, C.c_int, pi32, i32, pf, pu8, pi64]),
        'iss_wait': (C.c_int, [vp, i64]),
        'iss_comm_unique_id': (C.c_int, [vp, pu8]),
        'iss_comm_init': (C.c_int, [vp, pu8, i32, i32]),
        'iss_comm_destroy': (C.c_int, [vp]),
        'iss_allgather_segments': (C.c_int, [vp, pi32, i32, i32, pi32, pi32]),
        'iss_comm_allreduce_max': (C.c_int, [vp, pd]),
        'iss_comm_info': (C.c_int, [vp, pi32, pi32, pi32, C.c_char_p, i32]),
        'iss_sidekit': (C.c_int, [vp, pi32]),
        'iss_get_loge': (C.c_int, [vp, pf]),
        'iss_get_mspec': (C.c_int, [vp, pf]),
        'iss_set_mspec': (C.c_int, [vp, pf, i32]),
        'iss_cnn_load': (C.c_int, [vp, C.c_int, pi32, i32, pf, i64, i32, pi64, i32, i32, i32, i32]),
        'iss_cnn_probs': (C.c_int, [vp, C.c_int, pi32, i32, pf, pu8]),
        'iss_cnn_forward': (C.c_int, [vp, C.c_int, pf, i32, pf]),
        'iss_cnn_flops': (C.c_int, [vp, C.c_int, pd]),
        'iss_set_precision': (C.c_int, [vp, C.c_int]),
        'iss_set_precision_guard': (C.c_int, [vp, C.c_float]),
        'iss_cnn_precision_info': (C.c_int, [vp, C.c_int, pi32, pf, pi32, pi32, pf]),
        'iss_cnn_set_net_precision': (C.c_int, [vp, C.c_int, C.c_int]),
        'iss_vbx_tables': (C.c_int, [vp, pd, pd]),
        'iss_vbx_features': (C.c_int, [vp, pi32, pd, i64, pf, pi32]),
        'iss_vbx_set_dither': (C.c_int, [vp, pd, i64]),
        'iss_vbx_features_pcm16': (C.c_int, [vp, pi16, i64, pf, pi32]),
        'iss_vbx_embed': (C.c_int, [vp, C.c_int, pi32, i32, pf]),
        'iss_prof_enable': (C.c_int, [vp, C.c_int]),
        'iss_prof_get': (C.c_int, [vp, C.c_int, pd, pi64, pd]),
        'iss_prof_reset': (C.c_int, [vp]),
        'iss_prof_get_row': (C.c_int, [vp, C.c_int, pd, pi64]),
        'iss_prof_get_instance': (C.c_int, [vp, C.c_int, C.c_char_p, i32, pd, pi64, pd]),
        'iss_set_diag': (C.c_int, [vp, C.c_uint32]),
        'iss_viterbi_f64': (C.c_int, [pd, i64, i32, pd, pi32]),
        'iss_viterbi_f32': (C.c_int, [pf, i64, i32, pd, pi32]),
        'iss_energy_viterbi': (C.c_int, [pf, i64, C.c_double, C.c_double, C.c_double, pd, pi32]),
        'iss_viterbi_segments_f32': (C.c_int, [pf, pi64, i64, i32, pd, pi32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    L._iss_symbols = tuple(sig)
    _lib = L
    return L


DIAG_BITS = {'no_shared_first': 0x001, 'no_flrows': 0x002, 'no_ws': 0x004, 'no_ws3': 0x008, 'no_direct1': 0x010,
             'no_nh2': 0x020, 'no_tr': 0x040, 'no_pw': 0x080, 'no_pws': 0x100, 'no_pws2': 0x200, 'no_wq': 0x400,
             'no_dual': 0x800, 'no_chain': 0x1000, 'no_ring': 0x2000, 'no_fsame': 0x4000, 'no_f32ws': 0x8000, 'no_ncb1': 0x10000, 'no_wsu3': 0x20000, 'no_gfused': 0x40000, 'no_hl': 0x80000}                                                                                         # include/iss.h ISS_DIAG_*


def diag_flags(names):
    """'no_shared_first,no_pws2' -> ISS_DIAG_* bit mask (unknown names raise)."""
    flags = 0
    for nm in str(names).replace(' ', '').replace('+', ',').lower().split(','):
        if nm:
            flags |= DIAG_BITS[nm]
    return flags


def _ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def viterbi(emission, transition):
    """Host Viterbi (iss_viterbi_f32/f64): emission (T,K) float32|float64 log-scores,
    transition (K,K) float64.  Returns int32 state ids (T,)."""
    L = lib()
    em = np.ascontiguousarray(emission)
    if em.dtype not in (np.float32, np.float64):
        em = em.astype(np.float64)
    tr = np.ascontiguousarray(transition, dtype=np.float64)
    T, K = em.shape
    out = np.empty(T, dtype=np.int32)
    if em.dtype == np.float32:
        rc = L.iss_viterbi_f32(_ptr(em, C.c_float), T, K, _ptr(tr, C.c_double), _ptr(out, C.c_int32))
    else:
        rc = L.iss_viterbi_f64(_ptr(em, C.c_double), T, K, _ptr(tr, C.c_double), _ptr(out, C.c_int32))
    if rc != 0:
        raise NativeError(f"iss_viterbi failed ({rc}): T={T} K={K}")
    return out


def viterbi_segments(emission, seg_len, transition):
    """iss_viterbi_segments_f32: emission (n,K) float32 log-scores of consecutive segments of lengths seg_len, each smoothed on
    its own.  Returns int32 state ids (n,)."""
    L = lib()
    em = np.ascontiguousarray(emission, dtype=np.float32)
    sl = np.ascontiguousarray(seg_len, dtype=np.int64)
    tr = np.ascontiguousarray(transition, dtype=np.float64)
    n, K = em.shape
    assert int(sl.sum()) == n
    out = np.empty(n, dtype=np.int32)
    rc = L.iss_viterbi_segments_f32(_ptr(em, C.c_float), _ptr(sl, C.c_int64), sl.size, K, _ptr(tr, C.c_double), _ptr(out, C.c_int32))
    if rc != 0:
        raise NativeError(f"iss_viterbi_segments_f32 failed ({rc}): n={n} K={K} segments={sl.size}")
    return out


_LOG_EPS = np.log(np.full(1, 1e-10))[0], np.log(np.full(1, 1 - 1e-10))[0]     # as pred2logemission's np.log(ret) yields them


def energy_viterbi(loge, threshold, transition):
    """iss_energy_viterbi: smoothed activity (T,) int32 of `loge > threshold` (see include/iss.h)."""
    L = lib()
    x = np.ascontiguousarray(loge, dtype=np.float32)
    tr = np.ascontiguousarray(transition, dtype=np.float64)
    out = np.empty(x.size, dtype=np.int32)
    rc = L.iss_energy_viterbi(_ptr(x, C.c_float), x.size, float(threshold), float(_LOG_EPS[0]), float(_LOG_EPS[1]),
                              _ptr(tr, C.c_double), _ptr(out, C.c_int32))
    if rc != 0:
        raise NativeError(f"iss_energy_viterbi failed ({rc}): T={x.size}")
    return out


class Context:
    """One device context (stream + resident signal/features + loaded networks)."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.iss_create(int(device), C.byref(h))
        if rc != 0:
            raise NativeError(f"iss_create(device={device}) failed ({rc}): "
                              f"{self._L.iss_last_error(None).decode()}")
        self._h = h
        self.device = device
        self.T = 0
        self._net_out = {}
        self._dither_n = 0          # length of the dither stream cached on the device (vbx)
        self._pinned = {}           # data address -> hipHostMalloc pointer of pinned_empty() arrays
        self.comm_rank, self.comm_world = 0, 1
        self.precision = None       # last value given to set_precision / set_workspace_limit (None = library default)
        self.workspace_limit = None
        self.diag = 0
        env = os.environ.get('ISS_DIAG', '')
        if env:                     # A/B tooling only (tools/ab_env.sh): the library itself never reads the environment
            self.set_diag(diag_flags(env))
        self.guard_threshold = None # last value given to set_precision_guard (None = library default, 5e-4)
        env = os.environ.get('ISS_PREC_GUARD', '')
        if env:                     # A/B tooling only: timing-only experiment builds compute wrong results on purpose
            self.set_precision_guard(float(env))

    def close(self):
        if getattr(self, '_h', None):
            for p in list(self._pinned.values()):
                self._L.iss_host_free(self._h, C.c_void_p(p))
            self._pinned.clear()
            self._L.iss_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise NativeError(f"{what} failed ({rc}): {self._L.iss_last_error(self._h).decode()}")

    # ---- SIDEKIT front end
    def sidekit_tables(self, window, melbank):
        w = np.ascontiguousarray(window, dtype=np.float64)
        b = np.ascontiguousarray(melbank, dtype=np.float32)
        assert w.shape == (400,) and b.shape == (24, 257)
        self._ck(self._L.iss_sidekit_tables(self._h, _ptr(w, C.c_double), _ptr(b, C.c_float)), 'iss_sidekit_tables')

    def set_signal(self, sig):
        """sig: 1-D int16 (PCM) or float32 array, 16 kHz mono."""
        sig = np.ascontiguousarray(sig)
        if sig.dtype == np.int16:
            self._ck(self._L.iss_signal_pcm16(self._h, _ptr(sig, C.c_int16), sig.size), 'iss_signal_pcm16')
        elif sig.dtype == np.float32:
            self._ck(self._L.iss_signal_f32(self._h, _ptr(sig, C.c_float), sig.size), 'iss_signal_f32')
        else:
            raise TypeError(f"signal dtype {sig.dtype}: need int16 or float32")
        self._keep = sig            # the async H2D copy reads it until the next sync

    def set_signal_device(self, dev_ptr, n, producer_stream=None):
        """PCM16 samples already in this GPU's HBM.  The library's stream is made to wait for everything submitted so
        far to `producer_stream` (a hipStream_t handle as int, e.g. torch.cuda.current_stream().cuda_stream; None = the
        legacy default stream); see include/iss.h for the ordering contract."""
        self._ck(self._L.iss_signal_pcm16_device_stream(self._h, C.c_void_p(int(dev_ptr)), int(n),
                                                        C.c_void_p(int(producer_stream)) if producer_stream else None),
                 'iss_signal_pcm16_device')

    # ---- page-locked host arrays
    def pinned_empty(self, shape, dtype):
        """numpy array backed by hipHostMalloc memory (freed with the context, or by `pinned_free`)."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._ck(self._L.iss_host_alloc(self._h, n, C.byref(p)), 'iss_host_alloc')
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        self._pinned[a.ctypes.data] = p.value
        return a

    def pinned_free(self, a):
        p = self._pinned.pop(a.ctypes.data, None)
        if p is not None and self._h:
            self._ck(self._L.iss_host_free(self._h, C.c_void_p(p)), 'iss_host_free')

    def sidekit(self):
        t = C.c_int32()
        self._ck(self._L.iss_sidekit(self._h, C.byref(t)), 'iss_sidekit')
        self.T = t.value
        return self.T

    def get_loge(self):
        out = np.empty(self.T, dtype=np.float32)
        self._ck(self._L.iss_get_loge(self._h, _ptr(out, C.c_float)), 'iss_get_loge')
        return out

    def get_mspec(self):
        out = np.empty((self.T, 24), dtype=np.float32)
        self._ck(self._L.iss_get_mspec(self._h, _ptr(out, C.c_float)), 'iss_get_mspec')
        return out

    def set_mspec(self, mspec):
        m = np.ascontiguousarray(mspec, dtype=np.float32)
        assert m.ndim == 2 and m.shape[1] == 24
        self._ck(self._L.iss_set_mspec(self._h, _ptr(m, C.c_float), m.shape[0]), 'iss_set_mspec')
        self.T = m.shape[0]

    # ---- CNN engine
    def cnn_load(self, net_id, compiled):
        """compiled: keras_model.CompiledNet."""
        prog = np.ascontiguousarray(compiled.prog, dtype=np.int32)
        blob = np.ascontiguousarray(compiled.blob, dtype=np.float32)
        be = np.ascontiguousarray(compiled.buf_elems, dtype=np.int64)
        h, w, c = compiled.in_shape
        self._ck(self._L.iss_cnn_load(self._h, net_id, _ptr(prog, C.c_int32), prog.shape[0], _ptr(blob, C.c_float),
                                      blob.size, be.size, _ptr(be, C.c_int64), h, w, c, compiled.out_dim),
                 'iss_cnn_load')
        self._net_out[net_id] = compiled.out_dim

    def cnn_probs(self, net_id, win_row):
        wr = np.ascontiguousarray(win_row, dtype=np.int32)
        n = wr.size
        probs = np.empty((n, self._net_out[net_id]), dtype=np.float32)
        fin = np.empty(n, dtype=np.uint8)
        self._ck(self._L.iss_cnn_probs(self._h, net_id, _ptr(wr, C.c_int32), n, _ptr(probs, C.c_float),
                                       _ptr(fin, C.c_uint8)), 'iss_cnn_probs')
        return probs, fin.astype(bool)

    def cnn_probs_async(self, net_id, win_row, probs_out=None, finite_out=None):
        """Enqueue iss_cnn_probs and return (ticket, probs, finite_u8); the arrays are valid after wait(ticket).
        Pass pinned_empty() arrays as outputs if the D2H copy is to overlap host work."""
        wr = np.ascontiguousarray(win_row, dtype=np.int32)
        n = wr.size
        if probs_out is None:
            probs_out = np.empty((n, self._net_out[net_id]), dtype=np.float32)
        if finite_out is None:
            finite_out = np.empty(n, dtype=np.uint8)
        assert probs_out.size >= n * self._net_out[net_id] and finite_out.size >= n
        t = C.c_int64(-1)
        self._ck(self._L.iss_cnn_probs_async(self._h, net_id, _ptr(wr, C.c_int32), n, _ptr(probs_out, C.c_float),
                                             _ptr(finite_out, C.c_uint8), C.byref(t)), 'iss_cnn_probs_async')
        return t.value, probs_out, finite_out

    def wait(self, ticket=-1):
        self._ck(self._L.iss_wait(self._h, int(ticket)), 'iss_wait')

    # ---- multi-GPU exchange (RCCL)
    def comm_unique_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        self._ck(self._L.iss_comm_unique_id(self._h, _ptr(buf, C.c_uint8)), 'iss_comm_unique_id')
        return buf.tobytes()

    def comm_init(self, uid, rank, world):
        buf = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        assert buf.size == 128
        self._ck(self._L.iss_comm_init(self._h, _ptr(buf, C.c_uint8), int(rank), int(world)), 'iss_comm_init')
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        self._ck(self._L.iss_comm_destroy(self._h), 'iss_comm_destroy')
        self.comm_rank, self.comm_world = 0, 1

    def allgather_segments(self, rows, capacity):
        """rows: (k,4) int32 of this rank, or None = "my local work failed" -> ((world, capacity, 4) int32, counts (world,));
        rank r's first min(counts[r], capacity) rows are valid, counts[r] == -1 marks a rank that reported failure."""
        world = self.comm_world
        out = np.zeros((world, capacity, 4), dtype=np.int32)
        counts = np.zeros(world, dtype=np.int32)
        if rows is None:
            rows, n = np.zeros((1, 4), np.int32), -1
        else:
            rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 4)
            n = rows.shape[0]
        self._ck(self._L.iss_allgather_segments(self._h, _ptr(rows, C.c_int32), n, int(capacity),
                                                _ptr(out, C.c_int32), _ptr(counts, C.c_int32)), 'iss_allgather_segments')
        return out, counts

    def comm_info(self):
        """{'world', 'rank', 'version', 'lib'} as RCCL itself reports them for this context's communicator."""
        w, r, v = C.c_int32(), C.c_int32(), C.c_int32()
        buf = C.create_string_buffer(512)
        self._ck(self._L.iss_comm_info(self._h, C.byref(w), C.byref(r), C.byref(v), buf, 512), 'iss_comm_info')
        return {'world': w.value, 'rank': r.value, 'version': v.value, 'lib': buf.value.decode(errors='replace')}

    def comm_allreduce_max(self, value):
        v = C.c_double(float(value))
        self._ck(self._L.iss_comm_allreduce_max(self._h, C.byref(v)), 'iss_comm_allreduce_max')
        return v.value

    def cnn_forward(self, net_id, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        out = np.empty((n, self._net_out[net_id]), dtype=np.float32)
        self._ck(self._L.iss_cnn_forward(self._h, net_id, _ptr(x, C.c_float), n, _ptr(out, C.c_float)), 'iss_cnn_forward')
        return out

    def cnn_flops(self, net_id):
        f = C.c_double()
        self._ck(self._L.iss_cnn_flops(self._h, net_id, C.byref(f)), 'iss_cnn_flops')
        return f.value

    def set_precision(self, mode):
        """PREC_BF16X3 (default: split-bf16 MFMA, float32-class results) or PREC_F32 (exact f32 MFMA)."""
        self._ck(self._L.iss_set_precision(self._h, int(mode)), 'iss_set_precision')
        self.precision = int(mode)

    def set_precision_guard(self, threshold):
        """Precision guard (include/iss.h): max |d log p| between the split-operand and the exact-f32 arithmetic above which a
        patch network's first call switches it to the other split mode or to exact f32; <= 0 turns the probe off.  Default 5e-4."""
        self._ck(self._L.iss_set_precision_guard(self._h, float(threshold)), 'iss_set_precision_guard')
        self.guard_threshold = float(threshold)

    def cnn_precision_info(self, net_id):
        """{'mode': PREC_*, 'max_dlogp': probe figure or None, 'slots': windows compared, 'state': 'pending' | 'passed' |
        'escalated' | 'fixed'} of one loaded network."""
        mode, slots, state, d, du = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_float(0), C.c_float(0)
        self._ck(self._L.iss_cnn_precision_info(self._h, int(net_id), C.byref(mode), C.byref(d), C.byref(slots), C.byref(state), C.byref(du)),
                 'iss_cnn_precision_info')
        return {'mode': {PREC_F32: 'f32', PREC_F16X3: 'f16x3'}.get(mode.value, 'bf16x3'), 'max_dlogp': None if d.value < 0 else float(d.value),
                'max_dlogp_in_use': None if du.value < 0 else float(du.value),
                'slots': slots.value, 'state': ('pending', 'passed', 'escalated', 'fixed')[state.value]}

    def cnn_set_net_precision(self, net_id, mode):
        """Arithmetic of ONE network: PREC_BF16X3 / PREC_F32, or -1 to follow the context again (and be probed again)."""
        self._ck(self._L.iss_cnn_set_net_precision(self._h, int(net_id), int(mode)), 'iss_cnn_set_net_precision')

    def set_workspace_limit(self, nbytes):
        self._ck(self._L.iss_set_workspace_limit(self._h, int(nbytes)), 'iss_set_workspace_limit')
        self.workspace_limit = int(nbytes)

    def mirror_settings(self, other):
        """Give this context the arithmetic mode, workspace cap and profiling state of `other` (the extra device
        contexts of the multi-file pipeline follow the Segmenter's own context)."""
        if getattr(other, 'precision', None) is not None and other.precision != getattr(self, 'precision', None):
            self.set_precision(other.precision)
        if getattr(other, 'workspace_limit', None) is not None and other.workspace_limit != getattr(self, 'workspace_limit', None):
            self.set_workspace_limit(other.workspace_limit)
        if getattr(other, 'diag', 0) != getattr(self, 'diag', 0):
            self.set_diag(other.diag)
        if getattr(other, 'guard_threshold', None) is not None and other.guard_threshold != getattr(self, 'guard_threshold', None):
            self.set_precision_guard(other.guard_threshold)

    def set_diag(self, flags):
        """Kernel-selection switches (include/iss.h ISS_DIAG_*): an int, or names like 'no_shared_first,no_pws2'."""
        flags = diag_flags(flags) if isinstance(flags, str) else int(flags)
        self._ck(self._L.iss_set_diag(self._h, flags), 'iss_set_diag')
        self.diag = flags

    def synchronize(self):
        self._ck(self._L.iss_synchronize(self._h), 'iss_synchronize')

    # ---- VBx front end
    def vbx_tables(self, window, melbank):
        w = np.ascontiguousarray(window, dtype=np.float64)
        b = np.ascontiguousarray(melbank, dtype=np.float64)
        assert w.shape == (400,) and b.shape == (257, 64)
        self._ck(self._L.iss_vbx_tables(self._h, _ptr(w, C.c_double), _ptr(b, C.c_double)), 'iss_vbx_tables')

    def vbx_features(self, sig_i32, dither_u):
        s = np.ascontiguousarray(sig_i32, dtype=np.int32)
        u = np.ascontiguousarray(dither_u, dtype=np.float64)
        assert s.shape == u.shape and s.ndim == 1
        T = (s.size + 320 - 400) // 160 + 1
        out = np.empty((T, 64), dtype=np.float32)
        t = C.c_int32()
        self._ck(self._L.iss_vbx_features(self._h, _ptr(s, C.c_int32), _ptr(u, C.c_double), s.size,
                                          _ptr(out, C.c_float), C.byref(t)), 'iss_vbx_features')
        self._dither_n = 0          # this entry uploads its own stream over the cached one
        assert t.value == T
        return out

    def vbx_set_dither(self, dither_u):
        u = np.ascontiguousarray(dither_u, dtype=np.float64)
        self._ck(self._L.iss_vbx_set_dither(self._h, _ptr(u, C.c_double), u.size), 'iss_vbx_set_dither')
        self._dither_n = u.size

    def vbx_features_pcm16(self, pcm, to_host=True):
        """-> (T, 64) float32, or with to_host=False just T (the features stay on the device for iss_vbx_embed)."""
        s = np.ascontiguousarray(pcm, dtype=np.int16)
        T = (s.size + 320 - 400) // 160 + 1
        t = C.c_int32()
        if not to_host:
            self._ck(self._L.iss_vbx_features_pcm16(self._h, _ptr(s, C.c_int16), s.size, None, C.byref(t)), 'iss_vbx_features_pcm16')
            assert t.value == T
            return T
        out = np.empty((T, 64), dtype=np.float32)
        self._ck(self._L.iss_vbx_features_pcm16(self._h, _ptr(s, C.c_int16), s.size, _ptr(out, C.c_float), C.byref(t)),
                 'iss_vbx_features_pcm16')
        assert t.value == T
        return out

    def vbx_embed(self, net_id, starts):
        st = np.ascontiguousarray(starts, dtype=np.int32)
        out = np.empty((st.size, self._net_out[net_id]), dtype=np.float32)
        self._ck(self._L.iss_vbx_embed(self._h, net_id, _ptr(st, C.c_int32), st.size, _ptr(out, C.c_float)), 'iss_vbx_embed')
        return out

    # ---- profiling
    def prof_enable(self, on=True):
        self._ck(self._L.iss_prof_enable(self._h, 1 if on else 0), 'iss_prof_enable')

    def prof_reset(self):
        self._ck(self._L.iss_prof_reset(self._h), 'iss_prof_reset')

    def prof_get(self, kind):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._ck(self._L.iss_prof_get(self._h, kind, C.byref(ms), C.byref(n), C.byref(fl)), 'iss_prof_get')
        return ms.value, n.value, fl.value

    def prof_instances(self):
        """[{'kernel', 'ms', 'launches', 'flops'}] per distinct kernel instantiation launched since the last prof_reset."""
        out = []
        buf = C.create_string_buffer(192)
        i = 0
        while True:
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            if self._L.iss_prof_get_instance(self._h, i, buf, 192, C.byref(ms), C.byref(n), C.byref(fl)) != 0:
                break
            out.append({'kernel': buf.value.decode(), 'ms': ms.value, 'launches': n.value, 'flops': fl.value})
            i += 1
        return out

    def prof_get_row(self, row):
        """(ms, launches) of op-program row `row` since the last prof_reset (per-layer view of the same event brackets)."""
        ms, n = C.c_double(), C.c_int64()
        self._ck(self._L.iss_prof_get_row(self._h, int(row), C.byref(ms), C.byref(n)), 'iss_prof_get_row')
        return ms.value, n.value
