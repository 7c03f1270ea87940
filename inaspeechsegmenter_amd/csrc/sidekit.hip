// SIDEKIT log-mel front end for gfx950: 16 kHz PCM -> (log-energy, 24-band log-mel) per 10 ms frame.
//
// Replaces sidekit_mfcc.py:200-237 (power_spectrum), :240-275 (framing, pre_emphasis) and
// :325-334 (mel + log) of the reference as called by segmenter.py:58.  One fused kernel;
// nothing of the reference's (T,400)/(T,257) temporaries ever exists in HBM.
//
// Work decomposition: one 64-lane wavefront owns one frame at a time (4 frames per
// 256-thread workgroup, grid-stride over frames).
//   1. coalesced load of the frame's 400 samples (int16 -> x/32768 exactly, or f32),
//      per-frame pre-emphasis with a lane shuffle for x[i-1]  (float32, mul and sub rounded
//      separately like numpy: sidekit_mfcc.py:275)
//   2. log-energy: float32 sum of squares in numpy's pairwise order for n = 400
//      ( (P8[0:96]+P8[96:200]) + (P8[200:296]+P8[296:400]), P8 = 8 strided accumulators )
//      -> bit-identical partial sums to `(framed**2).sum(axis=1)` (:226)
//   3. Hann window in float64 (:223,231), 512-point real FFT in float64 as one 256-point
//      complex radix-4 DIF FFT held in LDS + real-input untangle, twiddles staged in LDS
//   4. |X|^2 in float64, stored as float32 (:233)
//   5. 24 triangular mel filters (454 non-zeros, LDS-resident rows, float32 FMA chains), log (:334); the log-energy and
//      the 24 bands share ONE float64 log call (lane 24 carries the energy)
// float64 is deliberate: the reference's FFT is float64 (window promotes) and the labels
// hinge on `loge > threshold`; this stage is ~4 GFLOP per audio-hour, nowhere near a bound.
#include "iss_internal.h"
#include "fft256.h"

// The reference rounds every float32 product and sum separately (numpy ufuncs); hipcc's default
// -ffp-contract=fast would fuse `x - 0.97*prev` and `acc + v*v` into FMAs and move the
// pre-emphasised samples by 1 ulp, which shows up as a -150 dB noise floor in the spectrum
// (6e-4 on log-mel values 18 nepers below the frame maximum).  HIP's __fmul_rn/__fsub_rn are plain
// operators inlined from a header, so the pragma alone is not enough: the Makefile also passes
// -ffp-contract=off (explicit fma()/fmaf() calls are still honoured).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float sample_at(const int16_t* p, int64_t i) { return (float)p[i] * (1.0f / 32768.0f); }
__device__ __forceinline__ float sample_at(const float* p, int64_t i) { return p[i]; }

// Every wavefront owns its frame's LDS slices (y, z, spec): the stages only need the wave's own LDS writes to have
// landed before its other lanes read them -- a wavefront-scope fence (s_waitcnt lgkmcnt(0)), not a workgroup barrier.
// (With __syncthreads between the six stages the four waves ran in lock step: 1.0 ms per audio-hour.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename SampleT>
__global__ __launch_bounds__(256) void sidekit_kernel(const SampleT* __restrict__ sig, int T,
                                                      const double* __restrict__ window,
                                                      const double* __restrict__ tw,
                                                      const float* __restrict__ melw,
                                                      const int32_t* __restrict__ mellim,
                                                      float* __restrict__ loge, float* __restrict__ mspec) {
    __shared__ cplx s_w256[256];
    __shared__ cplx s_w512[256];
    __shared__ double s_win[400];
    __shared__ float s_melw[1024];
    __shared__ int32_t s_lim[72];
    __shared__ cplx s_z[4][256];
    __shared__ float s_y[4][400];
    __shared__ float s_spec[4][256];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {
        const cplx* twc = reinterpret_cast<const cplx*>(tw);
        s_w256[tid] = twc[tid];
        s_w512[tid] = twc[256 + tid];
        for (int i = tid; i < 400; i += 256) s_win[i] = window[i];
        for (int i = tid; i < 1024; i += 256) s_melw[i] = melw[i];
        if (tid < 72) s_lim[tid] = mellim[tid];
    }
    __syncthreads();

    cplx* z = s_z[wv];
    float* y = s_y[wv];
    float* spec = s_spec[wv];
    const int frames_per_pass = gridDim.x * 4;
    const int npass = (T + frames_per_pass - 1) / frames_per_pass;
    // twiddles of this lane's butterflies in the span-64 and span-16 stages: constant across frames, kept in registers
    const Tw3 tw1 = bfly_twiddles(lane & 15, 4, s_w256), tw2 = bfly_twiddles(lane & 3, 16, s_w256);

    // the samples of the NEXT frame of this wave are fetched while the current one is transformed: the 7 loads of a frame
    // (64 x 2 bytes each) would otherwise expose one HBM round trip per frame in front of ~3 k cycles of arithmetic
    float xs[7];
    auto fetch = [&](int tt) {
        const int64_t s0 = (int64_t)(tt < T ? tt : T - 1) * 160;     // (a frame that does not exist re-reads the last one)
#pragma unroll
        for (int r = 0; r < 7; ++r) {
            const int i = lane + 64 * r;
            xs[r] = sample_at(sig, s0 + (i < 400 ? i : 399));
        }
    };
    fetch(blockIdx.x * 4 + wv);

    for (int pass = 0; pass < npass; ++pass) {
        const int t = pass * frames_per_pass + blockIdx.x * 4 + wv;
        const bool live = t < T;

        // ---- 1. per-frame pre-emphasis (sidekit_mfcc.py:275) on the prefetched samples; next frame's loads issued
        float xc[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) xc[r] = xs[r];
        if (pass + 1 < npass) fetch(t + frames_per_pass);
        if (live) {
            float carry = 0.f;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const int i = lane + 64 * r;
                float x = (i < 400) ? xc[r] : 0.f;
                float prev = __shfl_up(x, 1);
                if (lane == 0) prev = (r == 0) ? x : carry;
                carry = __shfl(x, 63);
                if (i < 400) y[i] = __fsub_rn(x, __fmul_rn(prev, 0.97f));
            }
        }
        wave_sync();

        // ---- 2. log-energy in numpy's pairwise order (sidekit_mfcc.py:226) -------------------
        // all 13 values of a lane's strided chain are read first (independent LDS reads), then summed in order; the sum
        // of squares is broadcast and its logarithm is taken in stage 5 by lane 24, in the SAME f64 log call as the 24
        // mel bands (the kernel is f64-VALU-issue-bound: one ~80-instruction log per frame instead of two)
        float esum = 0.f;
        if (live) {
            const int l = lane & 31, blk = l >> 3, j = l & 7;
            const int start = (blk == 0) ? 0 : (blk == 1) ? 96 : (blk == 2) ? 200 : 296;
            const bool longer = blk & 1;                              // 104 values instead of 96
            float v[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) v[i] = y[start + 8 * (i < 12 || longer ? i : 11) + j];
            float acc = __fmul_rn(v[0], v[0]);
#pragma unroll
            for (int i = 1; i < 12; ++i) acc = __fadd_rn(acc, __fmul_rn(v[i], v[i]));
            if (longer) acc = __fadd_rn(acc, __fmul_rn(v[12], v[12]));
            acc = __fadd_rn(acc, __shfl_xor(acc, 1));
            acc = __fadd_rn(acc, __shfl_xor(acc, 2));
            acc = __fadd_rn(acc, __shfl_xor(acc, 4));
            acc = __fadd_rn(acc, __shfl_xor(acc, 8));
            acc = __fadd_rn(acc, __shfl_xor(acc, 16));
            esum = __shfl(acc, 0);
        }

        // ---- 3. Hann (f64) + first radix-4 stage straight from y -----------------------------
        if (live) {
            cplx a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int n = lane + 64 * m;
                if (n < 200)
                    a[m] = make_double2((double)y[2 * n] * s_win[2 * n], (double)y[2 * n + 1] * s_win[2 * n + 1]);
                else
                    a[m] = make_double2(0.0, 0.0);
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

        // ---- 4. real-input untangle + power (sidekit_mfcc.py:232-233) -------------------------
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = lane + 64 * r;
                spec[k] = (float)untangle_power(z, k, s_w512);
            }
        }
        wave_sync();

        // ---- 5. mel bank + log (sidekit_mfcc.py:334), and the frame's log-energy ---------------
        if (live && lane < 25) {
            // lanes 0..23: one mel band each (up to 48 bins): the reference's `spec @ fbank.T` is a float32 product
            // (sgemm), so four independent float32 FMA chains over 8-bin groups (16 LDS reads in flight; accumulating in
            // float64 cost two v_cvt_f64_f32 per bin at the f64 issue rate).  Lane 24: the sum of squares of stage 2.
            // One float64 log for all 25 lanes: correctly rounded float32 results, like the reference's np.log on float32.
            float arg = esum;
            if (lane < 24) {
                const int lo = s_lim[lane * 3], nb = s_lim[lane * 3 + 1], off = s_lim[lane * 3 + 2];
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                int i = 0;
                for (; i + 8 <= nb; i += 8) {
                    float sp[8], w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { sp[q] = spec[lo + i + q]; w[q] = s_melw[off + i + q]; }
                    a0 = fmaf(sp[0], w[0], a0); a1 = fmaf(sp[1], w[1], a1); a2 = fmaf(sp[2], w[2], a2); a3 = fmaf(sp[3], w[3], a3);
                    a0 = fmaf(sp[4], w[4], a0); a1 = fmaf(sp[5], w[5], a1); a2 = fmaf(sp[6], w[6], a2); a3 = fmaf(sp[7], w[7], a3);
                }
                for (; i < nb; ++i) a0 = fmaf(spec[lo + i], s_melw[off + i], a0);
                arg = (a0 + a1) + (a2 + a3);
            }
            const float r = (float)log((double)arg);
            if (lane < 24) mspec[(size_t)t * 24 + lane] = r;
            else loge[t] = r;
        }
        wave_sync();
    }
}

}  // namespace

int iss_launch_sidekit(iss_ctx* c) {
    const int T = c->T;
    int blocks = (T + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    iss_prof_begin(c, 1, 0.0);
    if (c->sig_kind == 1)
        hipLaunchKernelGGL(sidekit_kernel<int16_t>, dim3(blocks), dim3(256), 0, c->stream,
                           (const int16_t*)c->sig_ptr, T, c->d_window, c->d_tw, c->d_melw, c->d_mellim,
                           (float*)c->loge.p, (float*)c->mspec.p);
    else
        hipLaunchKernelGGL(sidekit_kernel<float>, dim3(blocks), dim3(256), 0, c->stream,
                           (const float*)c->sig_ptr, T, c->d_window, c->d_tw, c->d_melw, c->d_mellim,
                           (float*)c->loge.p, (float*)c->mspec.p);
    iss_prof_end(c);
    ISS_HIP(c, hipGetLastError());
    return ISS_OK;
}
