// conv_x3_pwc_kernel: two CHAINED pointwise GEMMs per 128-row tile -- the 1x1 expansion of a Bottleneck with its identity
// residual and ReLU, and the 1x1 reduction of the NEXT Bottleneck that reads its result (resnet.py:48-75):
//
//     x' = relu(r . We^T + be + x)          (C1 -> C2 channels, in place over x)            row j of the op program
//     q  = act(x' . Wr^T + br)              (C2 -> 128 channels)                            row j + 1
//
// As two launches the C2-channel tensor x' (the largest of the block) is written by the first and read back by the second:
// per-layer counters (profiles/r04_vbx_layer_hbm.md) show every 1x1 layer of the x-vector path moving its algorithmic bytes
// once at 4.2-4.6 TB/s -- only removing a round trip helps.  Here the second GEMM consumes x' out of LDS:
//
//   per tile: the r rows are split into bf16 hi / lo once and stay in LDS (K1T k-tiles of 32 channels);
//   per step s = 32 output channels of the first GEMM = k-tile s of the second (C2 / 32 steps):
//     * We rows [32 s, +32) (all C1 k), Wr columns [32 s, +32) (all 128 rows) and the residual block x[:, 32 s .. +32) were
//       requested two steps ago into one of two register sets; the weights go to LDS between two barriers
//     * first GEMM, 32 x 32 per wave (transposed accumulators, two of them to halve the dependent MFMA chain)
//     * its epilogue through the wave-private LDS transpose: + bias + residual, relu, the x' block is STORED (full 128-byte
//       lines; the next block needs it as its residual) and split into the wave's rows of the second GEMM's A k-tile
//     * second GEMM step: 4 column blocks x 2 k16 x 3 MFMAs into the tile's 128 x 128 accumulators
//   after the last step: second epilogue (bias, activation) through the same staging.
//
// One workgroup (4 waves, 32 rows each) per CU: 155 KB of LDS.  Loads are compiler-visible (two register sets, a load cursor
// two steps ahead of the compute cursor that runs across tile boundaries, as in conv_x3_pw_kernel).
#pragma once
#include "conv_pw.h"

namespace issk {

#ifndef ISS_PWC_EXP                                  // timing-only experiment builds (wrong results), never set by the Makefile:
#define ISS_PWC_EXP 0                                // 1 no MFMAs, 2 no stores, 4 no barriers, 8 no residual loads, 16 no weight loads
#endif
constexpr bool PWC_X_NOMFMA = ISS_PWC_EXP & 1, PWC_X_NOST = ISS_PWC_EXP & 2, PWC_X_NOBAR = ISS_PWC_EXP & 4, PWC_X_NORES = ISS_PWC_EXP & 8,
               PWC_X_NOW = ISS_PWC_EXP & 16;

template <int K1T, int NC3>                          // C1 = 32 K1T input channels of the first GEMM, C3 = 32 NC3 outputs of the second
__global__ __launch_bounds__(256, 1) void conv_x3_pwc_kernel(const ConvArgs p) {
    constexpr int C1 = 32 * K1T;
    constexpr int C3 = 32 * NC3;
    constexpr int NWR = NC3 >= 2 ? NC3 / 2 : 1;      // 16-byte pieces of a Wr-slice plane per thread (C3 rows x 4 pieces)
    static_assert(NC3 == 1 || NC3 == 2 || NC3 == 4, "");
    constexpr int WLD = C1 + 8;                      // padded row of the We slice (bf16)
    constexpr int NWE = C1 >= 64 ? C1 / 64 : 1;      // 16-byte pieces of a We-slice plane per thread (32 rows x C1 / 8 pieces)
    __shared__ __attribute__((aligned(16))) uint16_t sRh[K1T][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sRl[K1T][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sWeh[32 * WLD];
    __shared__ __attribute__((aligned(16))) uint16_t sWel[32 * WLD];
    __shared__ __attribute__((aligned(16))) uint16_t sWrh[C3 * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sWrl[C3 * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sAh[BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sAl[BM * XLD];
    __shared__ __attribute__((aligned(16))) float sE[4 * 32 * PWS_ELD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ntiles = p.nblk;                  // row tiles: every workgroup computes all columns of both GEMMs
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    const int C2 = p.Cout;                           // channels of x / x' = K of the second GEMM
    const int nsteps = C2 / 32;
    const int k8 = tid & 7, lr = tid >> 3;           // r staging: k columns [4 k8, +4) of rows lr + 32 j
    const int br = tid >> 2, bseg = tid & 3;         // Wr staging: 8 bf16 of rows br and br + 64
    const int wer = tid >> 3, wes = tid & 7;         // We staging: row wer of the slice, 16-byte pieces wes (+ 8 i)
    const int li = lane & 31, lh = lane >> 5;
    const int er = lane >> 3, ec = (lane & 7) * 4;   // row-major side of the transposes: row er + 8 j, 4 channels from ec
    float* const E = sE + wv * 32 * PWS_ELD;

    // ---- operands of a step.  As in conv_x3_pws_kernel every load of the steady state is issued from inline asm into tied
    // registers and the kernel counts the vector-memory instructions it has issued (`issued`): hipcc's own waitcnt placement
    // drains a ring that lives across loop iterations (it waited with vmcnt(0) at the top of every step: 3.9 us per step
    // where the loads need 1.8).  The weight slices (L2-resident, one register set) are requested one step ahead, the
    // residual block + bias (HBM, two sets) two steps ahead; one mark, behind the weight request, covers both (the residual
    // of the same step is older).  Counted: the asm loads and the unpredicated stores of full tiles; anything else only makes
    // a wait longer than necessary.  Register budget: everything that is in flight must stay in the 256 architectural VGPRs
    // (hipcc parks surplus values in AGPRs with a copy right behind the load, i.e. before the data has arrived;
    // tools/check_ring_regs.py looks for exactly that), so the r rows of the next tile are NOT requested ahead.
    struct WSet { u32x4 weh[NWE], wel[NWE], wrh[NWR], wrl[NWR]; };
    struct XSet { f32x4 res[4]; f32x4 b1; };
    auto rows_left = [&](unsigned tt) { const long long m0 = (long long)tt * BM; return (int)(p.M - m0 < BM ? p.M - m0 : BM); };
    unsigned weo[NWE], wro[NWR];                     // per-lane byte offsets of the weight pieces (the same for every step)
    const bool wr_on = br < C3;                      // (C3 = 32: the first 128 threads carry the Wr slice)
#pragma unroll
    for (int i = 0; i < NWE; ++i) weo[i] = (unsigned)(wer * p.Kpad + ((C1 >= 64 || wes < C1 / 8) ? wes * 8 + 64 * i : 0)) * 2u;
#pragma unroll
    for (int i = 0; i < NWR; ++i) wro[i] = (unsigned)((wr_on ? br + 64 * i : 0) * C2 + bseg * 8) * 2u;
    const unsigned bo = (unsigned)ec * 4u;
    struct RowOff { unsigned x[4]; };                // per-lane byte offsets into a tile of x (residual / store layout)
    auto row_offsets = [&](unsigned tt) {
        RowOff o;
        const int left = rows_left(tt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wv * 32 + er + 8 * j;
            o.x[j] = (unsigned)((row < left ? row : left - 1) * C2 + ec) * 4u;
        }
        return o;
    };
    unsigned issued = 0, markW = 0, mark_r = 0;
    WSet W;
    XSet X0, X1;
#pragma unroll
    for (int i = 0; i < NWE; ++i) { W.weh[i] = W.wel[i] = u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int i = 0; i < NWR; ++i) { W.wrh[i] = W.wrl[i] = u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { X0.res[j] = X1.res[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    X0.b1 = X1.b1 = f32x4{0.f, 0.f, 0.f, 0.f};

    // residual cursor, two steps ahead of the compute cursor; past the last tile it re-reads the last one
    unsigned tl = t;
    int sl = 0;
    RowOff OL = row_offsets(t);
#define ISS_PWC_GATHER_W(SW)                      /* weight slices of step SW */                                    \
    {                                                                                                                \
        const uint16_t* eh_ = p.wh + (size_t)(32 * (SW)) * p.Kpad;                                                   \
        const uint16_t* el_ = p.wl + (size_t)(32 * (SW)) * p.Kpad;                                                   \
        const uint16_t* rh_ = p.wh2 + 32 * (SW);                                                                     \
        const uint16_t* rl_ = p.wl2 + 32 * (SW);                                                                     \
        if (!PWC_X_NOW) {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NWE; ++i) { ISS_PWS_LD(W.weh[i], weo[i], eh_); ISS_PWS_LD(W.wel[i], weo[i], el_); } \
        _Pragma("unroll") for (int i = 0; i < NWR; ++i) { ISS_PWS_LD(W.wrh[i], wro[i], rh_); ISS_PWS_LD(W.wrl[i], wro[i], rl_); } \
        issued += 2 * NWE + 2 * NWR; } markW = issued;                                                                     \
    }
#define ISS_PWC_GATHER_X(X)                       /* residual block and bias of the cursor's step; the cursor advances */ \
    {                                                                                                                \
        const float* xs_ = p.res + (size_t)tl * BM * C2 + 32 * sl;                                                   \
        const float* bs_ = p.bias + 32 * sl;                                                                         \
        if (!PWC_X_NORES) {                                                                                          \
        ISS_PWS_LD(X.res[0], OL.x[0], xs_); ISS_PWS_LD(X.res[1], OL.x[1], xs_);                                      \
        ISS_PWS_LD(X.res[2], OL.x[2], xs_); ISS_PWS_LD(X.res[3], OL.x[3], xs_);                                      \
        ISS_PWS_LD(X.b1, bo, bs_);                                                                                   \
        issued += 5; }                                                                                               \
        if (++sl == nsteps) {                                                                                        \
            sl = 0;                                                                                                  \
            tl = tl + gridDim.x < ntiles ? tl + gridDim.x : tl;                                                      \
            OL = row_offsets(tl);                                                                                    \
        }                                                                                                            \
    }
    // the r rows of tile TT: requested, waited for, split and staged (a tile boundary drains the ring once)
    auto load_r = [&](unsigned tt) __attribute__((always_inline)) {
        f32x4 ra[K1T][4];
        const int left = rows_left(tt);
        unsigned ro[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int rr = lr + 32 * j; ro[j] = (unsigned)((rr < left ? rr : left - 1) * C1 + k8 * 4) * 4u; }
        const float* rs_ = p.in + (size_t)tt * BM * C1;
#pragma unroll
        for (int kt = 0; kt < K1T; ++kt) {
            const float* rk_ = rs_ + kt * XBK;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ra[kt][j] = f32x4{0.f, 0.f, 0.f, 0.f}; ISS_PWS_LD(ra[kt][j], ro[j], rk_); }
        }
        issued += 4 * K1T; mark_r = issued;
        pws_wait_outstanding(issued - mark_r);
#pragma unroll
        for (int kt = 0; kt < K1T; ++kt) asm volatile("" : "+v"(ra[kt][0]), "+v"(ra[kt][1]), "+v"(ra[kt][2]), "+v"(ra[kt][3]));
#pragma unroll
        for (int kt = 0; kt < K1T; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 h, l;
                pws_split4(ra[kt][j], h, l);
                *reinterpret_cast<bf16x4*>(&sRh[kt][(lr + 32 * j) * XLD + k8 * 4]) = h;
                *reinterpret_cast<bf16x4*>(&sRl[kt][(lr + 32 * j) * XLD + k8 * 4]) = l;
            }
    };

    floatx16 acc2[NC3];
    f32x4 bias2[NC3];                                // second GEMM's bias in the store layout: the same for every tile
#pragma unroll
    for (int c = 0; c < NC3; ++c) {
        bias2[c] = *reinterpret_cast<const f32x4*>(p.bias2 + 32 * c + ec);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[c][i] = 0.f;
    }
    const int aoff = (wv * 32 + li) * XLD + lh * 8;  // A fragments: the wave's rows
    const int boff_s = li * XLD + lh * 8;            // B fragments of the second GEMM (Wr slice rows)
    const int weoff = li * WLD + lh * 8;             // B fragments of the first GEMM (We slice rows)

    RowOff OC = OL;                                  // offsets of the compute cursor's tile
    load_r(t);
    ISS_PWC_GATHER_X(X0)
    ISS_PWC_GATHER_X(X1)
    ISS_PWC_GATHER_W(0)
    int sc = 0;                                      // compute cursor: step of tile t
    bool done = false;
    bool full = rows_left(t) == BM;

    auto step_body = [&](XSet& X) __attribute__((always_inline)) {
        // ---- first GEMM: 32 channels [32 sc, +32) x the wave's 32 rows, K = C1
        floatx16 a1, b1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a1[i] = 0.f; b1[i] = 0.f; }
#pragma unroll
        for (int kt = 0; kt < K1T; ++kt) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 rh = *reinterpret_cast<const bf16x8*>(&sRh[kt][aoff + ks * 16]);
                const bf16x8 rl = *reinterpret_cast<const bf16x8*>(&sRl[kt][aoff + ks * 16]);
                const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&sWeh[weoff + kt * XBK + ks * 16]);
                const bf16x8 wl = *reinterpret_cast<const bf16x8*>(&sWel[weoff + kt * XBK + ks * 16]);
                if (PWC_X_NOMFMA) { a1[0] += (float)rh[0] + (float)wl[0]; b1[0] += (float)rl[0] + (float)wh[0]; }
                else if (ks == 0) {
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, rl, a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, rh, a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, rh, a1, 0, 0, 0);
                } else {
                    b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, rl, b1, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, rh, b1, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, rh, b1, 0, 0, 0);
                }
            }
        }
        // ---- its epilogue: transpose through the wave's staging rows, + bias + residual, relu, store x', split into sA
        {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v[0] = a1[4 * g + 0] + b1[4 * g + 0]; v[1] = a1[4 * g + 1] + b1[4 * g + 1];
                v[2] = a1[4 * g + 2] + b1[4 * g + 2]; v[3] = a1[4 * g + 3] + b1[4 * g + 3];
                *reinterpret_cast<f32x4*>(&E[li * PWS_ELD + 8 * g + 4 * lh]) = v;
            }
            float* const xo = p.out + (size_t)t * BM * C2 + 32 * sc;
            const int left = full ? BM : rows_left(t);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wv * 32 + er + 8 * j;
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(8 * j + er) * PWS_ELD + ec]);
                v = v + X.b1;
                v = v + X.res[j];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                if (PWC_X_NOST) { if (v[0] == 123.456f) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xo) + OC.x[j]) = v; }
                else if (full) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xo) + OC.x[j]) = v;
                else if (row < left) *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xo) + OC.x[j]) = v;
                bf16x4 h, l;
                pws_split4(v, h, l);
                *reinterpret_cast<bf16x4*>(&sAh[row * XLD + ec]) = h;
                *reinterpret_cast<bf16x4*>(&sAl[row * XLD + ec]) = l;
            }
            if (full && !PWC_X_NOST) issued += 4;
        }
        __builtin_amdgcn_sched_barrier(0);
        ISS_PWC_GATHER_X(X)                          // the set's registers are free: the residual of step + 2
        __builtin_amdgcn_sched_barrier(0);
        // ---- second GEMM, k-tile sc: the wave's rows of sA (written by this wave) x the Wr slice, two column blocks at a time
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sAh[aoff + ks * 16]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sAl[aoff + ks * 16]);
#pragma unroll
            for (int c0 = 0; c0 < NC3; c0 += 2) {
                constexpr int NB = NC3 >= 2 ? 2 : 1;
                bf16x8 bh[NB], bl[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    bh[c] = *reinterpret_cast<const bf16x8*>(&sWrh[boff_s + (c0 + c) * 32 * XLD + ks * 16]);
                    bl[c] = *reinterpret_cast<const bf16x8*>(&sWrl[boff_s + (c0 + c) * 32 * XLD + ks * 16]);
                }
                if (PWC_X_NOMFMA) { acc2[c0][0] += (float)bh[0][0] + (float)al[0] + (float)bl[0][0] + (float)ah[0]; continue; }
#pragma unroll
                for (int c = 0; c < NB; ++c) acc2[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[c], al, acc2[c0 + c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NB; ++c) acc2[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[c], ah, acc2[c0 + c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < NB; ++c) acc2[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[c], ah, acc2[c0 + c], 0, 0, 0);
            }
        }
        if (++sc == nsteps) {                        // tile complete: second epilogue (bias, activation), 32 channels at a time
            float* const qo = p.out2 + (size_t)t * BM * C3;
            const int left = full ? BM : rows_left(t);
#pragma unroll
            for (int c = 0; c < NC3; ++c) {
                const f32x4 bias = bias2[c];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    v[0] = acc2[c][4 * g + 0]; v[1] = acc2[c][4 * g + 1]; v[2] = acc2[c][4 * g + 2]; v[3] = acc2[c][4 * g + 3];
                    *reinterpret_cast<f32x4*>(&E[li * PWS_ELD + 8 * g + 4 * lh]) = v;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wv * 32 + er + 8 * j;
                    f32x4 v = *reinterpret_cast<const f32x4*>(&E[(8 * j + er) * PWS_ELD + ec]);
                    v = v + bias;
                    if (p.act2 == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    float* const dst = qo + (size_t)row * C3 + 32 * c + ec;
                    if (full) *reinterpret_cast<f32x4*>(dst) = v;
                    else if (row < left) *reinterpret_cast<f32x4*>(dst) = v;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[c][i] = 0.f;
            }
            if (full) issued += 4 * NC3;
            sc = 0;
            t += gridDim.x;
            if (t >= ntiles) done = true;
            else {
                full = rows_left(t) == BM;
                OC = row_offsets(t);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_s_barrier();        // every wave has read its last r fragments of the old tile
                load_r(t);
            }
        }
    };
    // one step; X: its residual set (requested two steps ago), reloaded for step + 2 inside step_body
#define ISS_PWC_STEP(X)                                                                                               \
    {                                                                                                                \
        __builtin_amdgcn_s_waitcnt(0xc07f);          /* lgkmcnt(0): this wave's LDS reads of the previous step */     \
        if (!PWC_X_NOBAR) __builtin_amdgcn_s_barrier();                /* every wave is done with the previous weight slices (tile start: sR complete) */ \
        pws_wait_outstanding(issued - markW);        /* this step's weight slices -- and its residual block, which is older */ \
        asm volatile("" : "+v"(X.res[0]), "+v"(X.res[1]), "+v"(X.res[2]), "+v"(X.res[3]), "+v"(X.b1));               \
        _Pragma("unroll") for (int i = 0; i < NWR; ++i) asm volatile("" : "+v"(W.wrh[i]), "+v"(W.wrl[i]));           \
        _Pragma("unroll") for (int i = 0; i < NWE; ++i) {                                                            \
            asm volatile("" : "+v"(W.weh[i]), "+v"(W.wel[i]));                                                       \
            if (C1 >= 64 || wes < C1 / 8) {                                                                          \
                *reinterpret_cast<u32x4*>(&sWeh[wer * WLD + wes * 8 + 64 * i]) = W.weh[i];                           \
                *reinterpret_cast<u32x4*>(&sWel[wer * WLD + wes * 8 + 64 * i]) = W.wel[i];                           \
            }                                                                                                        \
        }                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NWR; ++i) {                                                            \
            if (wr_on) {                                                                                             \
                *reinterpret_cast<u32x4*>(&sWrh[(br + 64 * i) * XLD + bseg * 8]) = W.wrh[i];                         \
                *reinterpret_cast<u32x4*>(&sWrl[(br + 64 * i) * XLD + bseg * 8]) = W.wrl[i];                         \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_s_waitcnt(0xc07f);                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        { const int sw_ = sc + 1 == nsteps ? 0 : sc + 1; ISS_PWC_GATHER_W(sw_) }       /* the next step's slices */  \
        if (!PWC_X_NOBAR) __builtin_amdgcn_s_barrier();                                                                              \
        step_body(X);                                                                                                \
    }
    while (true) {
        ISS_PWC_STEP(X0)
        if (done) break;
        ISS_PWC_STEP(X1)
        if (done) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISS_PWC_STEP
#undef ISS_PWC_GATHER_W
#undef ISS_PWC_GATHER_X
}

// host: rows j (expansion, in place, relu, residual) and j + 1 (reduction to 128 channels) of an op program as one launch.
// a: ConvArgs of row j with wh2 / wl2 / bias2 / out2 / act2 of row j + 1.
// (C1, C3) pairs the kernel is instantiated for: ResNet-101's stages 1-3 and the transitions between them (cnn_pwc.hip)
inline bool pwc_compiled(int c1, int c3) {
    return (c1 == 32 && (c3 == 32 || c3 == 64)) || (c1 == 64 && (c3 == 64 || c3 == 128)) || (c1 == 128 && c3 == 128);
}
inline bool pwc_supported(const ConvArgs& a) {
    return a.bias && a.bias2 && a.res && a.res == a.out && !a.ps && a.act == 1 && a.act2 <= 1 && a.pp == 1 && a.Kpad == a.Cin &&
           pwc_compiled(a.Cin, a.Cout2) && a.Cout % 32 == 0 && a.Cout >= 128 && a.Cout <= 2048 &&
           a.M * (long long)a.Cout < (1ll << 40);
}
void iss_pwc_launch(const ConvArgs& a, hipStream_t st);

}  // namespace issk
