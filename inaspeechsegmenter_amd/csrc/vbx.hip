// VBx / HTK-style 64-band log-mel front end for gfx950 (float64 like the reference).
//
// Replaces vbx_segmenter.py:72-89 `get_features`: add_dither (features_vbx.py:127-128) ->
// reflect pad 120/200 (vbx_segmenter.py:86) -> fbank_htk (features_vbx.py:62-120: frames
// 400/160, per-frame mean removal, per-frame pre-emphasis 0.97, Povey window, |rfft512|^2,
// mel matmul, log(max(1, .))) -> cmvn_floating_kaldi(150,149) (features_vbx.py:131-149).
//
// Three kernels:
//   vbx_fbank_kernel   one wavefront per frame (as the SIDEKIT kernel); the reflect padding
//                      and the dither add are index arithmetic in the loader, the frame mean
//                      uses numpy's pairwise order in float64 -> bit-identical to x.mean(1)
//   vbx_cumsum_kernel  ONE wavefront, lane = mel channel, sequential float64 running sum
//                      over time: np.cumsum(x, 0) is order-sensitive, a parallel scan would
//                      round differently (360 k dependent adds ~ 1-2 ms per audio-hour;
//                      the x-vector network behind it costs ~1 s, so exactness wins)
//   vbx_cmn_kernel     x - (f[ws+wl]-f[ws])/wl, cast to float32
#include "iss_internal.h"
#include "fft256.h"

namespace {

// wavefront-scope LDS hand-off (see sidekit.hip): each wave owns its frame's LDS slices
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename SampleT>
__device__ __forceinline__ double seg_at(const SampleT* __restrict__ sig, const double* __restrict__ u, int64_t n,
                                         int64_t j) {
    // seg = r_[x[119::-1], x, x[-1:-201:-1]]  with x = sig + 8*(u*2-1)
    int64_t i = j < 120 ? 119 - j : (j < 120 + n ? j - 120 : n - 1 - (j - 120 - n));
    const double d = __dmul_rn(8.0, __dadd_rn(__dmul_rn(u[i], 2.0), -1.0));
    return __dadd_rn((double)sig[i], d);
}

template <typename SampleT>
__global__ __launch_bounds__(256) void vbx_fbank_kernel(const SampleT* __restrict__ sig, const double* __restrict__ u,
                                                        int64_t n, int T, const double* __restrict__ window,
                                                        const double* __restrict__ tw, const double* __restrict__ melw,
                                                        const int32_t* __restrict__ mellim, double* __restrict__ fb) {
    __shared__ cplx s_w256[256];
    __shared__ cplx s_w512[256];
    __shared__ double s_win[400];
    __shared__ double s_melw[512];
    __shared__ int32_t s_lim[192];
    __shared__ cplx s_z[4][256];
    __shared__ double s_x[4][400];
    __shared__ double s_spec[4][256];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const cplx* twc = reinterpret_cast<const cplx*>(tw);
        s_w256[tid] = twc[tid];
        s_w512[tid] = twc[256 + tid];
        for (int i = tid; i < 400; i += 256) s_win[i] = window[i];
        for (int i = tid; i < 512; i += 256) s_melw[i] = melw[i];
        if (tid < 192) s_lim[tid] = mellim[tid];
    }
    __syncthreads();

    cplx* z = s_z[wv];
    double* x = s_x[wv];
    double* spec = s_spec[wv];
    const int frames_per_pass = gridDim.x * 4;
    const int npass = (T + frames_per_pass - 1) / frames_per_pass;
    const Tw3 tw1 = bfly_twiddles(lane & 15, 4, s_w256), tw2 = bfly_twiddles(lane & 3, 16, s_w256);   // see sidekit.hip
    // next frame's samples (+ dither) are fetched while the current frame is transformed (see sidekit.hip)
    double xs[7];
    auto fetch = [&](int tt) {
        const int64_t s0 = (int64_t)(tt < T ? tt : T - 1) * 160;
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int i = lane + 64 * r;
            xs[r] = seg_at(sig, u, n, s0 + (i < 400 ? i : 399));
        }
    };
    fetch(blockIdx.x * 4 + wv);

    for (int pass = 0; pass < npass; ++pass) {
        const int t = pass * frames_per_pass + blockIdx.x * 4 + wv;
        const bool live = t < T;
        if (live) {
#pragma unroll
            for (int r = 0; r < 7; ++r) { const int i = lane + 64 * r; if (i < 400) x[i] = xs[r]; }
        }
        if (pass + 1 < npass) fetch(t + frames_per_pass);
        wave_sync();
        // frame mean, numpy pairwise order (features_vbx.py:100-101)
        double mean = 0.0;
        if (live) {
            const int l = lane & 31, blk = l >> 3, j = l & 7;
            const int start = (blk == 0) ? 0 : (blk == 1) ? 96 : (blk == 2) ? 200 : 296;
            const int len = (blk & 1) ? 104 : 96;
            double acc = x[start + j];
            for (int i = 8; i < len; i += 8) acc = __dadd_rn(acc, x[start + i + j]);
            acc = __dadd_rn(acc, __shfl_xor(acc, 1));
            acc = __dadd_rn(acc, __shfl_xor(acc, 2));
            acc = __dadd_rn(acc, __shfl_xor(acc, 4));
            acc = __dadd_rn(acc, __shfl_xor(acc, 8));
            acc = __dadd_rn(acc, __shfl_xor(acc, 16));
            mean = __shfl(acc, 0) / 400.0;
        }
        // mean removal + pre-emphasis + window + first FFT stage (features_vbx.py:100-106)
        if (live) {
            cplx a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int q = lane + 64 * m;
                if (q < 200) {
                    const int i0 = 2 * q, i1 = 2 * q + 1;
                    const double c0 = __dadd_rn(x[i0], -mean), c1 = __dadd_rn(x[i1], -mean);
                    const double p0 = i0 == 0 ? c0 : __dadd_rn(x[i0 - 1], -mean);
                    const double y0 = __dadd_rn(c0, -__dmul_rn(p0, 0.97));
                    const double y1 = __dadd_rn(c1, -__dmul_rn(c0, 0.97));
                    a[m] = make_double2(__dmul_rn(y0, s_win[i0]), __dmul_rn(y1, s_win[i1]));
                } else {
                    a[m] = make_double2(0.0, 0.0);
                }
            }
            fft256_stage0(z, lane, a, s_w256);
        }
        wave_sync();
        if (live) bfly4<true>(z, (lane >> 4) * 64, 16, lane & 15, tw1);
        wave_sync();
        if (live) bfly4<true>(z, (lane >> 2) * 16, 4, lane & 3, tw2);
        wave_sync();
        if (live) bfly4<false>(z, lane * 4, 1, 0, tw2);
        wave_sync();
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                spec[k] = untangle_power(z, k, s_w512);
            }
        }
        wave_sync();
        if (live) {   // lane = mel channel (64 of them), features_vbx.py:113
            const int lo = s_lim[lane * 3], nb = s_lim[lane * 3 + 1], off = s_lim[lane * 3 + 2];
            double acc = 0.0;
            for (int i = 0; i < nb; ++i) acc += spec[lo + i] * s_melw[off + i];
            fb[(size_t)t * 64 + lane] = log(fmax(1.0, acc));
        }
        wave_sync();
    }
}

// f[0] = 0, f[t+1] = f[t] + x[t]   (np.r_[zeros, np.cumsum(x, 0)], features_vbx.py:145)
// np.cumsum's order is sequential in time, so the sum is ONE dependent chain of f64 adds per channel: one wavefront
// (lane = channel).  What must not sit on that chain is memory latency: rows are fetched in register blocks of 32, the
// block after the current one is always in flight (two named blocks, loop unrolled by two, no branch between a block's
// loads and the adds that hide them), so a row costs its v_add_f64 + store issue instead of 1/8 of an HBM round trip
// (the round-2 kernel: 8 loads, wait, 8 adds -- 62 ns per row, 22 ms per audio-hour, 87 % of the feature stage).
__global__ __launch_bounds__(64) void vbx_cumsum_kernel(const double* __restrict__ fb, int T, double* __restrict__ f) {
    constexpr int B = 32;
    const int lane = threadIdx.x;
    double acc = 0.0;
    f[lane] = 0.0;
    int t = 0;
    const double* src = fb + lane;
    double* dst = f + 64 + lane;
    if (T >= 2 * B) {
        double va[B], vb[B];
        const long long last = (long long)(T - B) * 64;              // first row of the last whole block that exists
#pragma unroll
        for (int q = 0; q < B; ++q) va[q] = src[(size_t)q * 64];
        // invariant at the top: va = rows [t, t + B), and t + 2 B <= T
        while (true) {
#pragma unroll
            for (int q = 0; q < B; ++q) vb[q] = src[(size_t)(t + B + q) * 64];
#pragma unroll
            for (int q = 0; q < B; ++q) { acc = __dadd_rn(acc, va[q]); dst[(size_t)(t + q) * 64] = acc; }
            const bool more = t + 4 * B <= T;
            {   // rows [t + 2 B, t + 3 B) for the next round (clamped to rows that exist: read, not used, on the last one)
                long long o = (long long)(t + 2 * B) * 64;
                o = o < last ? o : last;
#pragma unroll
                for (int q = 0; q < B; ++q) va[q] = src[o + q * 64];
            }
#pragma unroll
            for (int q = 0; q < B; ++q) { acc = __dadd_rn(acc, vb[q]); dst[(size_t)(t + B + q) * 64] = acc; }
            t += 2 * B;
            if (!more) break;
        }
    }
    for (; t + 8 <= T; t += 8) {                                     // < 4 B rows left
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(t + q) * 64];
#pragma unroll
        for (int q = 0; q < 8; ++q) { acc = __dadd_rn(acc, v[q]); dst[(size_t)(t + q) * 64] = acc; }
    }
    for (; t < T; ++t) { acc = __dadd_rn(acc, src[(size_t)t * 64]); dst[(size_t)t * 64] = acc; }
}

__global__ void vbx_cmn_kernel(const double* __restrict__ fb, const double* __restrict__ f, int T, int LC, int win_len,
                               float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)T * 64) return;
    const int t = (int)(idx >> 6), ch = (int)(idx & 63);
    int ws = t - LC;                                     // features_vbx.py:144
    if (ws > T - win_len) ws = T - win_len;
    if (ws < 0) ws = 0;
    const double m = __dadd_rn(f[(size_t)(ws + win_len) * 64 + ch], -f[(size_t)ws * 64 + ch]) / (double)win_len;
    out[idx] = (float)__dadd_rn(fb[idx], -m);
}

}  // namespace

extern "C" int iss_vbx_tables(iss_ctx* c, const double* window400, const double* bank /* (257,64) */) {
    if (!c || !window400 || !bank) return iss_fail(c, ISS_EINVAL, "iss_vbx_tables: NULL argument");
    ISS_HIP(c, hipSetDevice(c->device));
    std::vector<int32_t> lim(64 * 3);
    std::vector<double> w;
    for (int ch = 0; ch < 64; ++ch) {
        int lo = -1, hi = -1;
        for (int k = 0; k < 257; ++k)
            if (bank[k * 64 + ch] != 0.0) { if (lo < 0) lo = k; hi = k; }
        if (lo < 0) { lo = 0; hi = -1; }
        if (hi >= 256) return iss_fail(c, ISS_EINVAL, "vbx mel channel %d touches bin 256; unsupported table", ch);
        lim[ch * 3] = lo; lim[ch * 3 + 1] = hi - lo + 1; lim[ch * 3 + 2] = (int32_t)w.size();
        for (int k = lo; k <= hi; ++k) w.push_back(bank[k * 64 + ch]);
    }
    if (w.size() > 512) return iss_fail(c, ISS_EINVAL, "vbx mel bank has %zu weights (>512)", w.size());
    w.resize(512, 0.0);
    std::vector<double> tw(1024);
    fft256_host_twiddles(tw.data());
    if (!c->d_vbx_window) ISS_HIP(c, hipMalloc((void**)&c->d_vbx_window, 400 * sizeof(double)));
    if (!c->d_vbx_melw) ISS_HIP(c, hipMalloc((void**)&c->d_vbx_melw, 512 * sizeof(double)));
    if (!c->d_vbx_mellim) ISS_HIP(c, hipMalloc((void**)&c->d_vbx_mellim, 192 * sizeof(int32_t)));
    if (!c->d_tw) ISS_HIP(c, hipMalloc((void**)&c->d_tw, tw.size() * sizeof(double)));
    ISS_HIP(c, hipMemcpy(c->d_vbx_window, window400, 400 * sizeof(double), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_vbx_melw, w.data(), 512 * sizeof(double), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_vbx_mellim, lim.data(), 192 * sizeof(int32_t), hipMemcpyHostToDevice));
    ISS_HIP(c, hipMemcpy(c->d_tw, tw.data(), tw.size() * sizeof(double), hipMemcpyHostToDevice));
    c->vbx_tables = true;
    return ISS_OK;
}

namespace {
// common tail: signal (int32 or int16, already on the device) + dither stream (device) -> resident (T,64) float32
template <typename SampleT>
int vbx_run(iss_ctx* c, const SampleT* d_sig, const double* d_u, int64_t n, float* out, int32_t* T_out) {
    const int64_t T64 = (n + 320 - 400) / 160 + 1;
    if (T64 <= 0 || T64 > (1 << 26)) return iss_fail(c, ISS_EINVAL, "iss_vbx_features: unsupported length");
    const int T = (int)T64;
    int rc;
    if ((rc = iss_reserve(c, c->vbx_fb, (size_t)(2 * T + 1) * 64 * 8))) return rc;
    if ((rc = iss_reserve(c, c->vbx_out, (size_t)T * 64 * 4))) return rc;
    double* fb = (double*)c->vbx_fb.p;
    double* f = fb + (size_t)T * 64;
    int blocks = (T + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    iss_prof_begin(c, 1, 0.0);
    hipLaunchKernelGGL(vbx_fbank_kernel<SampleT>, dim3(blocks), dim3(256), 0, c->stream, d_sig, d_u, n, T, c->d_vbx_window,
                       c->d_tw, c->d_vbx_melw, c->d_vbx_mellim, fb);
    iss_prof_end(c);
    iss_prof_begin(c, 2, 0.0);
    hipLaunchKernelGGL(vbx_cumsum_kernel, dim3(1), dim3(64), 0, c->stream, fb, T, f);
    const int win_len = T < 300 ? T : 300;           // min(len(x), LC+RC+1), features_vbx.py:143
    hipLaunchKernelGGL(vbx_cmn_kernel, dim3((unsigned)(((long long)T * 64 + 255) / 256)), dim3(256), 0, c->stream, fb, f,
                       T, 150, win_len, (float*)c->vbx_out.p);
    iss_prof_end(c);
    ISS_HIP(c, hipGetLastError());
    if (out) ISS_HIP(c, hipMemcpyAsync(out, c->vbx_out.p, (size_t)T * 64 * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    c->vbx_T = T;
    if (T_out) *T_out = T;
    return ISS_OK;
}
}  // namespace

extern "C" int iss_vbx_features(iss_ctx* c, const int32_t* sig, const double* u, int64_t n, float* out, int32_t* T_out) {
    if (!c || !sig || !u || n < 200) return iss_fail(c, ISS_EINVAL, "iss_vbx_features: bad argument (need n >= 200)");
    if (!c->vbx_tables) return iss_fail(c, ISS_ESTATE, "iss_vbx_features: call iss_vbx_tables first");
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = iss_reserve(c, c->vbx_sig, (size_t)n * 4))) return rc;
    if ((rc = iss_reserve(c, c->vbx_dither, (size_t)n * 8))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->vbx_sig.p, sig, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    ISS_HIP(c, hipMemcpyAsync(c->vbx_dither.p, u, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    c->vbx_dither_n = 0;                              // the cached stream (iss_vbx_set_dither) was overwritten
    return vbx_run<int32_t>(c, (const int32_t*)c->vbx_sig.p, (const double*)c->vbx_dither.p, n, out, T_out);
}

extern "C" int iss_vbx_set_dither(iss_ctx* c, const double* u, int64_t n) {
    if (!c || !u || n <= 0) return iss_fail(c, ISS_EINVAL, "iss_vbx_set_dither: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = iss_reserve(c, c->vbx_dither, (size_t)n * 8))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->vbx_dither.p, u, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    c->vbx_dither_n = n;
    return ISS_OK;
}

extern "C" int iss_vbx_features_pcm16(iss_ctx* c, const int16_t* pcm, int64_t n, float* out, int32_t* T_out) {
    if (!c || !pcm || n < 200) return iss_fail(c, ISS_EINVAL, "iss_vbx_features_pcm16: bad argument (need n >= 200)");
    if (!c->vbx_tables) return iss_fail(c, ISS_ESTATE, "iss_vbx_features_pcm16: call iss_vbx_tables first");
    if (c->vbx_dither_n < n)
        return iss_fail(c, ISS_ESTATE, "iss_vbx_features_pcm16: cached dither stream holds %lld values, need %lld (iss_vbx_set_dither)",
                        (long long)c->vbx_dither_n, (long long)n);
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = iss_reserve(c, c->vbx_sig, (size_t)n * 2))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->vbx_sig.p, pcm, (size_t)n * 2, hipMemcpyHostToDevice, c->stream));
    return vbx_run<int16_t>(c, (const int16_t*)c->vbx_sig.p, (const double*)c->vbx_dither.p, n, out, T_out);
}
