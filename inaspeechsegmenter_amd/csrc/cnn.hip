// Small-CNN / ResNet engine for gfx950: executes a flat op program (include/iss.h) on
// batches of 20 ms slots (or x-vector windows).
//
// Replaces, for the reference: segmenter.py:76-88 `_get_patches` (never materialised here:
// the 68 x nmel window is gathered straight from the resident (T,24) log-mel inside the
// first conv's operand loader, z-normalised on the fly), segmenter.py:156-163 (gather +
// `keras predict`) and vbx_segmenter.py:262-266 (onnxruntime ResNet-101, resnet.py:78-135).
//
// Kernels
//   patch_stats_kernel   per-slot mean / population-std / finite flag     (segmenter.py:82,86)
//   conv_x3_kernel       conv2d + dense as implicit GEMM on v_mfma_f32_32x32x16_bf16 with both
//                        operands split into bf16 hi + lo and three MFMAs per k-step
//                        (hi*hi + hi*lo + lo*hi, f32 accumulate): 2^-16 relative operand error,
//                        i.e. float32-class results at 16/3 x the f32-MFMA rate (gfx950 has no
//                        xf32/TF32).  Default.
//   conv_x3_pw_kernel    1x1 stride-1 convolutions / dense layers: persistent GEMM, prefetch two k-tiles ahead
//   conv_igemm_kernel    the same GEMM on v_mfma_f32_32x32x2_f32 (exact f32, 157 TFLOP/s peak);
//                        iss_set_precision(ctx, ISS_PREC_F32)
//   both: fused bias, residual add, activation, post-activation scale/shift (BatchNorm),
//         optional fused non-overlapping max/avg pooling of 2 or 4 outputs, NHWC in/out
//   pool_kernel          max / average pooling, NHWC (pools that cannot be fused)
//   softmax_kernel       softmax over channels
//   statpool_kernel      mean || std over time                           (resnet.py:123-127)
#include <map>
#include "conv_common.h"
#include "conv_ws.h"
#include "conv_wq.h"
#include "conv_wq3.h"
#include "conv_wq3h.h"
#include "conv_dhl.h"
#include "conv_pwc.h"
#include "conv_pw.h"

using namespace issk;

namespace {

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution, exact f32.  C[m][n] = sum_k A[m][k] * Wt[n][k],
//   m = GEMM row (map_row), n = cout, k = (ky, kx, cin)   (NHWC, weights [Cout][Kpad]).
// 256 threads = 4 wavefronts; wave w owns rows [32w, 32w+32) x 64 cols = two 32x32 MFMA tiles.
// MODE: 0 = NHWC with Cin % 4 == 0 (float4 gathers), 1 = NHWC scalar gathers, 2 = z-normed patch.
template <int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned mt, nt;
    gemm_tile_of_block(blockIdx.x, p.nblk, p.nblk_n, mt, nt);
    const long long m0 = (long long)mt * BM;
    const int n0 = (int)nt * BN;

    // this thread fills (row lr, k4) and (row lr+64, k4) of sA
    const int k4 = tid & 3;
    const int lr = tid >> 2;
    RowSrc rs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) rs[j] = row_source<MODE>(p, m0 + lr + 64 * j);
    const int bn = n0 + lr;                       // weight row this thread stages
    const bool bok = bn < p.Cout;
    const float* wrow = p.w + (size_t)(bok ? bn : 0) * p.Kpad + k4 * 4;

    float4 ra[2], rb;
    auto gather = [&](int kt) {
        const int kbase = kt * BK + k4 * 4;
        if (MODE == 0) {
            const int2 e = reinterpret_cast<const int2*>(p.ktab)[kbase];
            const int ky = e.y >> 16, kx = e.y & 0xffff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int iy = rs[j].iy0 + ky, ix = rs[j].ix0 + kx;
                ra[j] = ld4_or_zero(p.in, rs[j].base + e.x,
                                    rs[j].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
                ra[j] = make_float4(gather_scalar<MODE>(p, rs[j], kbase), gather_scalar<MODE>(p, rs[j], kbase + 1),
                                    gather_scalar<MODE>(p, rs[j], kbase + 2), gather_scalar<MODE>(p, rs[j], kbase + 3));
        }
        rb = *reinterpret_cast<const float4*>(wrow + (size_t)kt * BK);        // rows >= Cout read row 0: never stored
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&sA[buf][lr * LDK + k4 * 4]) = ra[0];
        *reinterpret_cast<float4*>(&sA[buf][(lr + 64) * LDK + k4 * 4]) = ra[1];
        *reinterpret_cast<float4*>(&sB[buf][lr * LDK + k4 * 4]) = rb;
    };

    floatx16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int nk = p.Kpad / BK;
    gather(0);
    stage(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gather(kt + 1);
        const float* a_s = &sA[cur][(wv * 32 + li) * LDK + lh * 4];
        const float* b_s = &sB[cur][li * LDK + lh * 4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(a_s + g * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(b_s + g * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(b_s + 32 * LDK + g * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }
    epilogue_tile(p, acc0, m0 + wv * 32, n0 + li, lh);
    epilogue_tile(p, acc1, m0 + wv * 32, n0 + 32 + li, lh);
}

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution on bf16 MFMA with split operands ("bf16x3").
//   x = hi + lo,  hi = bf16(x),  lo = bf16(x - hi)   (|x - hi - lo| <= 2^-17 |x|)
//   a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi        (dropped a_lo*b_lo <= 2^-16 |a b|)
// Weights are split once on the host side of iss_cnn_load; activations stay f32 in HBM and
// are split while they are staged into LDS (v_cvt_pk_bf16_f32).  Same 128 x 64 tile / wave
// layout / epilogue as the f32 kernel, k-tile 32 = two v_mfma_f32_32x32x16_bf16 steps, i.e.
// 12 MFMAs per wave per k-tile.  LDS: 2 stages x (A hi+lo 20 KB + B hi+lo 10 KB) = 60 KB
// -> two workgroups per CU, one staging while the other is on the matrix pipe.
// MODE: 0 = NHWC with Cin % 32 == 0 (a k-tile is 32 consecutive channels of ONE tap: float4
//           gathers, no table), 1 = NHWC scalar gathers, 2 = z-normed patch.

// NTN = 32-column accumulator tiles per wavefront: the workgroup covers 128 rows x 32*NTN output channels.  NTN = 2 is
// the 128 x 64 tile of the other kernels; NTN = 4 / 8 (Cout >= 128 / 256) stage and split the A tile once for 128 / 256
// channels -- the ResNet 1x1 expansions (K = 32..256, N up to 1024) were bound by re-reading A once per 64 channels.
template <int MODE, bool TR, int NTN>
__global__ __launch_bounds__(256, NTN <= 4 ? 2 : 1) void conv_x3_kernel(const ConvArgs p) {
    constexpr int BNX = 32 * NTN;                    // output channels per workgroup
    constexpr int NBR = BNX / 64;                    // weight rows per thread and k-tile (rows br + 64 j)
    __shared__ __attribute__((aligned(16))) uint16_t sAh[2][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sAl[2][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBh[2][BNX * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBl[2][BNX * XLD];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned mt, nt;
    gemm_tile_of_block(blockIdx.x, p.nblk, p.nblk_n, mt, nt);
    const long long m0 = (long long)mt * BM;
    const int n0 = (int)nt * BNX;

    // A staging: thread fills k columns [4*k8, 4*k8+4) of rows lr, lr+32, lr+64, lr+96
    const int k8 = tid & 7;
    const int lr = tid >> 3;
    RowSrc rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rs[j] = row_source<MODE>(p, m0 + lr + 32 * j);
    // B staging: thread fills 8 bf16 (16 B) of weight rows br + 64 j, for hi and lo.  Rows >= Cout read row 0:
    // their output columns are never stored.
    const int br = tid >> 2, bseg = tid & 3;
    size_t boff[NBR];
#pragma unroll
    for (int j = 0; j < NBR; ++j) boff[j] = (size_t)(n0 + br + 64 * j < p.Cout ? n0 + br + 64 * j : 0) * p.Kpad + bseg * 8;

    float4 ra[4];
    uint4 rbh[NBR], rbl[NBR];
    int tap_ky = 0, tap_kx = 0, tap_c = 0;          // MODE 0 / 3: tap walked by the NEXT gather
    // MODE 3 (shared first layer, see row_source): the first layer's weight sums / bias of the 4 channels of this k-tile and which of
    // the 4 rows hold data (a zero-padded tap is a zero of the ACTIVATION, not of the raw row)
    float4 f_s4 = make_float4(0.f, 0.f, 0.f, 0.f), f_b4 = f_s4;
    float4 f_sx[4];                                  // MODE 4: S[x][c] of each row's tap column (the weight sum depends on the column)
    unsigned f_valid = 0;
    const float f_lob = p.f_act == 1 ? 0.f : -INFINITY;
    auto gather = [&](int kt) {
        if (MODE == 4) {
            // zero-padded ('same') first layer, see conv_ws.h FS: row iy of window b is shared row (win_row[b] - rmin) + iy, except the first
            // f_padt / last f_padb rows, which are the window's own edge rows f_erow0 + b * ne + e; the shift uses S[ix][c]
            const int cofs = tap_c + k8 * 4;
            f_b4 = *reinterpret_cast<const float4*>(p.f_bias + cofs);
            f_valid = 0;
            const int ne = p.f_padt + p.f_padb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = rs[j].iy0 + tap_ky, ix = rs[j].ix0 + tap_kx;
                const bool okj = rs[j].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int yb = iy - (p.H - p.f_padb);
                const int e = iy < p.f_padt ? iy : (yb >= 0 ? p.f_padt + yb : -1);
                const long long row = e >= 0 ? (long long)p.f_erow0 + (long long)rs[j].b * ne + e : rs[j].base + iy;
                ra[j] = ld4_or_zero(p.in, (row * p.W + ix) * p.Cin + cofs, okj);
                f_sx[j] = *reinterpret_cast<const float4*>(p.f_wsum + (okj ? ix * p.Cin + cofs : cofs));
                f_valid |= okj ? 1u << j : 0u;
            }
            tap_c += XBK;
            if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.kw) { tap_kx = 0; ++tap_ky; } }
        } else
        if (MODE == 0 || MODE == 3) {
            const int off = (tap_ky * p.W + tap_kx) * p.Cin + tap_c + k8 * 4;
            if (MODE == 3) {
                f_s4 = *reinterpret_cast<const float4*>(p.f_wsum + tap_c + k8 * 4);
                f_b4 = *reinterpret_cast<const float4*>(p.f_bias + tap_c + k8 * 4);
                f_valid = 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iy = rs[j].iy0 + tap_ky, ix = rs[j].ix0 + tap_kx;
                const bool okj = rs[j].ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                ra[j] = ld4_or_zero(p.in, rs[j].base + off, okj);
                if (MODE == 3) f_valid |= okj ? 1u << j : 0u;
            }
            tap_c += XBK;
            if (tap_c >= p.Cin) { tap_c = 0; if (++tap_kx == p.kw) { tap_kx = 0; ++tap_ky; } }
        } else {
            const int kbase = kt * XBK + k8 * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                ra[j] = make_float4(gather_scalar<MODE>(p, rs[j], kbase), gather_scalar<MODE>(p, rs[j], kbase + 1),
                                    gather_scalar<MODE>(p, rs[j], kbase + 2), gather_scalar<MODE>(p, rs[j], kbase + 3));
        }
#pragma unroll
        for (int j = 0; j < NBR; ++j) {
            rbh[j] = *reinterpret_cast<const uint4*>(p.wh + boff[j] + (size_t)kt * XBK);
            rbl[j] = *reinterpret_cast<const uint4*>(p.wl + boff[j] + (size_t)kt * XBK);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 h, l;
            if (MODE == 3 || MODE == 4) {                // the window's affine map + activation of the first layer, then the split
                const float sc = rs[j].sd, mr = rs[j].mean;
                if (MODE == 4) f_s4 = f_sx[j];
                float4 v = ra[j];
                v.x = fmaxf(fmaf(v.x, sc, fmaf(f_s4.x, mr, f_b4.x)), f_lob); v.y = fmaxf(fmaf(v.y, sc, fmaf(f_s4.y, mr, f_b4.y)), f_lob);
                v.z = fmaxf(fmaf(v.z, sc, fmaf(f_s4.z, mr, f_b4.z)), f_lob); v.w = fmaxf(fmaf(v.w, sc, fmaf(f_s4.w, mr, f_b4.w)), f_lob);
                ra[j] = (f_valid >> j) & 1u ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            split4(ra[j], h, l);
            *reinterpret_cast<bf16x4*>(&sAh[buf][(lr + 32 * j) * XLD + k8 * 4]) = h;
            *reinterpret_cast<bf16x4*>(&sAl[buf][(lr + 32 * j) * XLD + k8 * 4]) = l;
        }
#pragma unroll
        for (int j = 0; j < NBR; ++j) {
            *reinterpret_cast<uint4*>(&sBh[buf][(br + 64 * j) * XLD + bseg * 8]) = rbh[j];
            *reinterpret_cast<uint4*>(&sBl[buf][(br + 64 * j) * XLD + bseg * 8]) = rbl[j];
        }
    };

    floatx16 acc[NTN];
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int nk = p.Kpad / XBK;
    gather(0);
    stage(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    const int aoff = (wv * 32 + li) * XLD + lh * 8;
    const int boff_s = li * XLD + lh * 8;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gather(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sAh[cur][aoff + ks * 16]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sAl[cur][aoff + ks * 16]);
#pragma unroll
            for (int t = 0; t < NTN; ++t) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sBh[cur][boff_s + t * 32 * XLD + ks * 16]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sBl[cur][boff_s + t * 32 * XLD + ks * 16]);
                if (TR) {                                // C^T: rows = channels, columns = pixels (see epilogue_tr)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, al, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, ah, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, acc[t], 0, 0, 0);
                } else {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }
    if (TR) {
#pragma unroll
        for (int t = 0; t < NTN; t += 2) epilogue_tr(p, acc[t], acc[t + 1], m0 + wv * 32 + li, n0 + 32 * t, lh);
    } else {
#pragma unroll
        for (int t = 0; t < NTN; ++t) epilogue_tile(p, acc[t], m0 + wv * 32, n0 + 32 * t + li, lh);
    }
}

// Pointwise (1x1, stride 1, unpadded) convolutions = plain GEMMs on the NHWC pixel list: half of ResNet-101's flops in
// layers with K = 64 .. 512, i.e. 2 - 16 k-tiles per 128 x 64 output tile.  With one tile per workgroup (conv_x3_kernel)
// the first global load of every workgroup (~2 us) and its epilogue are exposed: rocprofv3 --pmc showed those launches at
// 2 TB/s of HBM traffic and 10 % of the matrix peak -- latency-bound, not bandwidth-bound.  Here 2 x 256 persistent
// workgroups walk the XCD-aware tile order and the register prefetch runs ACROSS tiles (the next tile's first k-tile is
// fetched during the current tile's last one, its stores drain behind the next tile's MFMAs); no row decomposition
// (row m of the GEMM is pixel m).
// Round 6: THREE workgroups per CU.  The segmenter nets' first dense layer (K = 4992 / 8320, 192 columns) has 660 tiles per launch;
// on 512 resident workgroups that was two rounds for 1.29 rounds of work, and a timing-only build whose activation loads all hit L2
// ran no faster (profiles/HISTORY.md, round 6: the layer is bound by its per-k-tile barrier structure, not by memory).  768 resident
// workgroups hold every tile at once.  LDS: the 80-byte padded rows (61 KB per workgroup) became unpadded 64-byte rows with the
// 16-byte chunk swizzle of conv_fp.h (chunk c of row r at c ^ ((r >> 2) & 3): the 16 lanes of a ds_read_b128 group hit 16
// different bank groups) = 48 KB.
constexpr int PWLD = XBK;                            // bf16 / fp16 elements per LDS row (64 bytes, swizzled)
template <bool F16>                                // fp16 instead of bf16 operand halves (ISS_PREC_F16X3, conv_common.h)
__global__ __launch_bounds__(256, 3) void conv_x3_pw_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) uint16_t sAh[2][BM * PWLD];
    __shared__ __attribute__((aligned(16))) uint16_t sAl[2][BM * PWLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBh[2][BN * PWLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBl[2][BN * PWLD];
    auto lds_off = [](int row, int chunk) { return row * PWLD + ((chunk ^ ((row >> 2) & 3)) << 3); };     // in 16-bit elements

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned ntiles = p.nblk * p.nblk_n;
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    const int k8 = tid & 7, lr = tid >> 3;           // A staging: k columns [4 k8, 4 k8 + 4) of rows lr + 32 j
    const int br = tid >> 2, bseg = tid & 3;         // B staging: 8 bf16 of weight row br, hi and lo
    const int awr = lds_off(lr, k8 >> 1) + (k8 & 1) * 4, bwr = lds_off(br, bseg);
    // A tile of the GEMM: uniform 64-bit base of its first row + 32-bit per-lane offsets (rows >= M re-read row M - 1:
    // their accumulator rows are never stored).  No masks and no branches around the loads: a select or a branch right
    // behind a load makes hipcc wait for it on the spot, in front of the MFMAs it is meant to overlap.
    struct Tile { long long m0; int n0; const float* abase; unsigned ao[4]; unsigned wo; };
    auto coords = [&](unsigned tt) {
        Tile T;
        unsigned mt, nt;
        gemm_tile_of_block(tt, p.nblk, p.nblk_n, mt, nt);
        T.m0 = (long long)mt * BM;
        T.n0 = (int)nt * BN;
        T.abase = p.in + T.m0 * p.Cin;
        const int left = (int)(p.M - T.m0 < BM ? p.M - T.m0 : BM);            // rows of this tile that exist
#pragma unroll
        for (int j = 0; j < 4; ++j) T.ao[j] = (unsigned)((lr + 32 * j < left ? lr + 32 * j : left - 1) * p.Cin + k8 * 4);
        T.wo = (unsigned)((T.n0 + br < p.Cout ? T.n0 + br : 0) * p.Kpad + bseg * 8);
        return T;
    };
    // Two register sets: a k-tile's loads are issued two iterations before they are converted into LDS.  One iteration is
    // only 12 MFMAs (~0.2 us) against ~2 us of memory latency, so what bounds these layers is the number of bytes in
    // flight; with a one-iteration distance the launches ran at 2 TB/s and 10 % of the matrix peak.
    struct Regs { float4 a[4]; uint4 bh, bl; };
    auto gather = [&](Regs& r, const Tile& T, int kt) {
#ifdef ISS_EXPERIMENTS                                           // timing-only (make EXPERIMENTS=1, ISS_DBG=4): every A load from one 64 KB region
        if (p.dbg & 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) r.a[j] = *reinterpret_cast<const float4*>(p.in + ((T.ao[j] + (unsigned)(kt * XBK)) & 0x3FFCu));
        } else
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) r.a[j] = *reinterpret_cast<const float4*>(T.abase + (T.ao[j] + (unsigned)(kt * XBK)));
        r.bh = *reinterpret_cast<const uint4*>(p.wh + (T.wo + (unsigned)(kt * XBK)));
        r.bl = *reinterpret_cast<const uint4*>(p.wl + (T.wo + (unsigned)(kt * XBK)));
    };
    auto stage = [&](Regs& r, int buf) {
        // opaque pass-through: the conversion can only start here, behind the MFMAs of the iteration
        asm volatile("" : "+v"(r.a[0].x), "+v"(r.a[0].y), "+v"(r.a[0].z), "+v"(r.a[0].w), "+v"(r.a[1].x), "+v"(r.a[1].y), "+v"(r.a[1].z),
                          "+v"(r.a[1].w), "+v"(r.a[2].x), "+v"(r.a[2].y), "+v"(r.a[2].z), "+v"(r.a[2].w), "+v"(r.a[3].x), "+v"(r.a[3].y),
                          "+v"(r.a[3].z), "+v"(r.a[3].w));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (F16) {
                uint2 h, l;
                split4_pk<true>(r.a[j], h.x, h.y, l.x, l.y);
                *reinterpret_cast<uint2*>(&sAh[buf][awr + 32 * j * PWLD]) = h;
                *reinterpret_cast<uint2*>(&sAl[buf][awr + 32 * j * PWLD]) = l;
            } else {
            bf16x4 h, l;
            split4(r.a[j], h, l);
            *reinterpret_cast<bf16x4*>(&sAh[buf][awr + 32 * j * PWLD]) = h;
            *reinterpret_cast<bf16x4*>(&sAl[buf][awr + 32 * j * PWLD]) = l;
            }
        }
        *reinterpret_cast<uint4*>(&sBh[buf][bwr]) = r.bh;
        *reinterpret_cast<uint4*>(&sBl[buf][bwr]) = r.bl;
    };

    floatx16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int nk = p.Kpad / XBK;
    const int li = lane & 31, lh = lane >> 5;
    // (rows lr + 32 j share (row >> 2) & 3, so one swizzled offset serves the four staging stores; the second k16 step of a row is
    // its first one's chunk ^ 2 = 16 elements further or nearer)
    const int aoff = lds_off(wv * 32 + li, lh);
    const int boff_s = lds_off(li, lh);

    // load cursor (tile, k-tile), two iterations ahead of the compute cursor; past the last tile it re-reads that tile
    unsigned tl = t;
    Tile TL = coords(t);
    int ktl = 0;
    auto advance_load = [&]() {
        if (++ktl == nk) {
            ktl = 0;
            tl = tl + gridDim.x < ntiles ? tl + gridDim.x : tl;
            TL = coords(tl);
        }
    };
    Tile TC = TL;                                    // compute cursor
    int ktc = 0;
    Regs r0, r1;
    gather(r0, TL, ktl); advance_load();
    gather(r1, TL, ktl); advance_load();
    stage(r0, 0);
    __syncthreads();
    int cur = 0;
    bool done = false;
    auto step = [&](Regs& rload, Regs& rstage) {     // one k-tile: loads for +2, MFMAs on `cur`, convert +1 into the other buffer
        gather(rload, TL, ktl);
        advance_load();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sAh[cur][aoff ^ (ks * 16)]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sAl[cur][aoff ^ (ks * 16)]);
            const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(&sBh[cur][boff_s ^ (ks * 16)]);
            const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(&sBl[cur][boff_s ^ (ks * 16)]);
            const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(&sBh[cur][(boff_s ^ (ks * 16)) + 32 * PWLD]);
            const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(&sBl[cur][(boff_s ^ (ks * 16)) + 32 * PWLD]);
            // C^T: rows = channels, columns = pixels (epilogue_tr); two independent accumulators alternate
            acc0 = mfma_x3<F16>(b0h, al, acc0);
            acc1 = mfma_x3<F16>(b1h, al, acc1);
            acc0 = mfma_x3<F16>(b0l, ah, acc0);
            acc1 = mfma_x3<F16>(b1l, ah, acc1);
            acc0 = mfma_x3<F16>(b0h, ah, acc0);
            acc1 = mfma_x3<F16>(b1h, ah, acc1);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stage(rstage, cur ^ 1);
        __syncthreads();
        cur ^= 1;
        if (++ktc == nk) {                           // tile complete
            epilogue_tr<true>(p, acc0, acc1, TC.m0 + wv * 32 + li, TC.n0, lh);
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            ktc = 0;
            t += gridDim.x;
            if (t >= ntiles) done = true; else TC = coords(t);
        }
    };
    while (true) {
        step(r0, r1);                                // r0 was staged last: free to load; r1 holds the next k-tile
        if (done) break;
        step(r1, r0);
        if (done) break;
    }
}

// ------------------------------------------------------------------------------------------
// First layer on the z-normalised 68 x h log-mel window (PATCH input, Cin = 1, kh*kw <= 32), bf16x3.
//
// conv1 is 1.8 % of the flops but its output is the largest tensor of the net (283-333 KB per slot),
// so the layer is bound by its stores; run through the generic kernel (table-driven scalar gathers into
// an LDS A tile, a barrier per k-tile) it took 18 % of the step at 3.8 % MFMA-busy.  Here:
//   * K fits ONE 32-wide k-tile, so each lane builds its own MFMA A fragments in registers -- 16 cached
//     loads from the (T,24) log-mel (a 128-row tile touches ~11 rows of one window), (x - mean) * (1/std),
//     bf16 hi/lo split -- no LDS A tile, no barrier at all;
//   * the workgroup's 64 x 32 weight block (hi + lo) is loaded once into 32 VGPRs and stays there while
//     the persistent workgroup walks its range of tiles: 12 MFMAs + the fused epilogue per tile.
// TR: operands swapped (C^T = W . A^T): a lane then holds 4 CONSECUTIVE channels of one pixel per register group and
// the epilogue stores float4 (8 x 16-byte stores per lane and tile instead of 32 x 4-byte ones); needs pp == 1.
template <bool TR>
__global__ __launch_bounds__(256, 2) void conv1_patch_x3_kernel(const ConvArgs p) {
    __shared__ int s_delta[XBK];                     // k -> ty * 24 + tx  (or -1 for the K padding)
    __shared__ int s_tap[XBK];                       // k -> (ty << 16) | tx
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int M = (int)p.M;
    if (tid < XBK) {
        const int K = p.H_k * p.kw;
        const int ty = tid / p.kw, tx = tid - ty * p.kw;
        s_delta[tid] = tid < K ? ty * 24 + tx : -1;
        s_tap[tid] = (ty << 16) | tx;
    }
    // weight fragments of this workgroup's 64 columns: B[k][n] at lane (n = li, k = 8 lh + e), k16 step ks
    bf16x8 b0h[2], b0l[2], b1h[2], b1l[2];
    {
        const int r0 = n0 + li < p.Cout ? n0 + li : 0, r1 = n0 + 32 + li < p.Cout ? n0 + 32 + li : 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            b0h[ks] = *reinterpret_cast<const bf16x8*>(p.wh + (size_t)r0 * p.Kpad + ks * 16 + lh * 8);
            b0l[ks] = *reinterpret_cast<const bf16x8*>(p.wl + (size_t)r0 * p.Kpad + ks * 16 + lh * 8);
            b1h[ks] = *reinterpret_cast<const bf16x8*>(p.wh + (size_t)r1 * p.Kpad + ks * 16 + lh * 8);
            b1l[ks] = *reinterpret_cast<const bf16x8*>(p.wl + (size_t)r1 * p.Kpad + ks * 16 + lh * 8);
        }
    }
    __syncthreads();
    int dl[16], tp[16];                              // this lane's 16 k values: k = ks * 16 + lh * 8 + e
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int k = (q >> 3) * 16 + lh * 8 + (q & 7);
        dl[q] = s_delta[k];
        tp[q] = s_tap[k];
    }
    const int per = ((int)p.nblk + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile_end = ((int)blockIdx.x + 1) * per < (int)p.nblk ? ((int)blockIdx.x + 1) * per : (int)p.nblk;
    for (int tile = (int)blockIdx.x * per; tile < tile_end; ++tile) {
        const int m = tile * BM + wv * 32 + li;
        int b, oy, ox;
        map_row32(p, m < M ? m : 0, b, oy, ox);
        const int iy0 = oy * p.sh - p.pt_, ix0 = ox * p.sw - p.pl_;
        const float* src = p.in + (size_t)p.win_row[b] * 24 + iy0 * 24 + ix0;
        const float mean = p.stats[2 * b];
        const float rsd = 1.0f / p.stats[2 * b + 1];
        const bool live = m < M && p.finite[b];
        float x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int iy = iy0 + (tp[q] >> 16), ix = ix0 + (tp[q] & 0xffff);
            const bool ok = live && dl[q] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float v = src[ok ? dl[q] : 0 - iy0 * 24 - ix0];       // row 0 of the window when masked: always mapped
            x[q] = ok ? (v - mean) * rsd : 0.f;
        }
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = x[ks * 8 + e];
                const __bf16 h = (__bf16)v;
                ah[ks][e] = h;
                al[ks][e] = (__bf16)(v - (float)h);
            }
        }
        floatx16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        if (!TR) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], b0h[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], b1h[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], b0l[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], b1l[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], b0h[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], b1h[ks], acc1, 0, 0, 0);
            }
            epilogue_tile(p, acc0, (long long)tile * BM + wv * 32, n0 + li, lh);
            epilogue_tile(p, acc1, (long long)tile * BM + wv * 32, n0 + 32 + li, lh);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {             // rows = channels (weight fragment), columns = pixels
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0h[ks], al[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1h[ks], al[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0l[ks], ah[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1l[ks], ah[ks], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0h[ks], ah[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1h[ks], ah[ks], acc1, 0, 0, 0);
            }
            epilogue_tr(p, acc0, acc1, m, n0, lh);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Per-slot statistics of the 68 x h log-mel window: mean, population std, finite flag.
// Shared first layer (ConvArgs::f_*): the PATCH conv without its per-window normalisation, once per log-mel row:
//   R[t][x][c] = sum_{ky,kx} w[c][ky * kw + kx] * mspec[(row0 + t + ky) * 24 + x + kx]       (f32, FMA chain in k order)
// One thread per (t, x, 4 channels), grid-stride (the weights are transposed into LDS once per workgroup).  Non-finite results are stored as 0: they only ever reach windows whose finite
// flag is 0, and those are scaled by 0 (their normalised input is all zeros in the reference, segmenter.py:86-88).
template <int KH_, int KW_>                       // compile-time filter shape (0, 0: run-time kh, kw)
__global__ __launch_bounds__(256) void first_layer_raw_kernel(const float* __restrict__ mspec, int row0, long long total,
                                                              int Wout, int Cout, int kh_, int kw_,
                                                              const float* __restrict__ w, int Kpad, float* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) float sW[];     // [K][Cout]: a lane's 4 channels of tap k are one ds_read_b128
    const int kh = KH_ ? KH_ : kh_, kw = KW_ ? KW_ : kw_;
    const int K = kh * kw;
    for (int e = threadIdx.x; e < K * Cout; e += 256) {
        const int co = e / K, k = e - co * K;
        sW[k * Cout + co] = w[(size_t)co * Kpad + k];
    }
    __syncthreads();
    const unsigned cg = (unsigned)Cout >> 2, utotal = (unsigned)total;       // total < 2^30 (host: 32-bit offsets into R)
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < utotal; idx += gridDim.x * 256u) {
        const unsigned pix = idx / cg;
        const int c4 = (int)(idx - pix * cg) * 4;
        const unsigned t = pix / (unsigned)Wout;
        const int x = (int)(pix - t * (unsigned)Wout);
        const float* src = mspec + (size_t)(row0 + (int)t) * 24 + x;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        auto tap = [&](int ky, int kx) {
            const float v = src[ky * 24 + kx];
            const float4 wk = *reinterpret_cast<const float4*>(&sW[(ky * kw + kx) * Cout + c4]);
            a0 = fmaf(v, wk.x, a0);
            a1 = fmaf(v, wk.y, a1);
            a2 = fmaf(v, wk.z, a2);
            a3 = fmaf(v, wk.w, a3);
        };
        if constexpr (KH_ > 0) {
#pragma unroll
            for (int ky = 0; ky < KH_; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW_; ++kx) tap(ky, kx);
        } else {
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) tap(ky, kx);
        }
        float4 o;
        o.x = isfinite(a0) ? a0 : 0.f; o.y = isfinite(a1) ? a1 : 0.f; o.z = isfinite(a2) ? a2 : 0.f; o.w = isfinite(a3) ? a3 : 0.f;
        *reinterpret_cast<float4*>(R + (size_t)idx * 4) = o;
    }
}

// A first layer with a fused non-overlapping MAX pool (conv - relu - pool - conv nets), shared between windows: max-pool commutes with the
// per-window map relu(R / std + t) (std > 0), so the pooled per-window activation is the map applied to maxpool(R).  A window's pool
// windows start at its own first row: window rows wr with (wr - rmin) % ph == q pool the pairs of PHASE q, so the pooled rows are kept
// in ph planes, plane q holding max over rows ph * r + q + {0..ph-1}, columns pw * x + {0..pw-1} of R; window b then reads plane q_b from
// row (wr_b - rmin - q_b) / ph on -- winrow_pool_kernel writes that as one row index into the stacked planes, so the consumer
// (conv_x3_kernel<3>) needs nothing but the transformed window list.  One thread per (plane, r, x, 4 channels).
__global__ __launch_bounds__(256) void pool_rows_kernel(const float* __restrict__ R, float* __restrict__ P, long long total, int rrows, int W,
                                                        int C, int ph, int pw, int plane_rows) {
    const unsigned cg = (unsigned)C >> 2, Wp = (unsigned)(W / pw);
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const unsigned c4 = (unsigned)(idx % cg) * 4u;
        long long t = idx / cg;
        const unsigned x = (unsigned)(t % Wp); t /= Wp;
        const int r = (int)(t % plane_rows), q = (int)(t / plane_rows);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < ph; ++dy) {
            int row = ph * r + q + dy;
            row = row < rrows ? row : rrows - 1;                     // (beyond the last row: never read by a window)
            for (int dx = 0; dx < pw; ++dx) {
                const float4 v = *reinterpret_cast<const float4*>(R + ((size_t)row * W + (x * pw + dx)) * C + c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(P + (size_t)idx * 4) = m;
    }
}
__global__ void winrow_pool_kernel(const int32_t* __restrict__ win_row, int32_t* __restrict__ out, int n, int rmin, int ph, int plane_rows) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int d = win_row[b] - rmin, q = d % ph;
    out[b] = (d - q) / ph + q * plane_rows;
}

// Zero-padded ('same') first layer, the rows every window shares (conv_x3_ws_kernel<..., FS>): output row t of the recording,
// column x, from ALL kh filter rows (input rows t - pt .. t - pt + kh - 1; a window reads row wr + y only for pt <= y < H - pb,
// where these are its own rows) and the filter columns that see data at x (0 <= x - pl + kx < W: the padding of the normalised
// window is worth the window mean in raw units, which the consumer accounts for through S[x][c]).  Rows outside the recording
// are clamped (never read).  One thread per (t, x, 4 channels), run-time filter shape.
__global__ __launch_bounds__(256) void first_layer_same_kernel(const float* __restrict__ mspec, int nrows_mspec, int row0, long long total,
                                                               int W, int Cout, int kh, int kw, int pt, int pl,
                                                               const float* __restrict__ w, int Kpad, float* __restrict__ R) {
    extern __shared__ __attribute__((aligned(16))) float sW[];     // [K][Cout]
    const int K = kh * kw;
    for (int e = threadIdx.x; e < K * Cout; e += 256) {
        const int co = e / K, k = e - co * K;
        sW[k * Cout + co] = w[(size_t)co * Kpad + k];
    }
    __syncthreads();
    const unsigned cg = (unsigned)Cout >> 2, utotal = (unsigned)total;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < utotal; idx += gridDim.x * 256u) {
        const unsigned pix = idx / cg;
        const int c4 = (int)(idx - pix * cg) * 4;
        const unsigned t = pix / (unsigned)W;
        const int x = (int)(pix - t * (unsigned)W);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            int r = row0 + (int)t - pt + ky;
            r = r < 0 ? 0 : (r > nrows_mspec - 1 ? nrows_mspec - 1 : r);
            const float* src = mspec + (size_t)r * 24;
            for (int kx = 0; kx < kw; ++kx) {
                const int ix = x - pl + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float v = src[ix];
                const float4 wk = *reinterpret_cast<const float4*>(&sW[(ky * kw + kx) * Cout + c4]);
                a0 = fmaf(v, wk.x, a0); a1 = fmaf(v, wk.y, a1); a2 = fmaf(v, wk.z, a2); a3 = fmaf(v, wk.w, a3);
            }
        }
        float4 o;
        o.x = isfinite(a0) ? a0 : 0.f; o.y = isfinite(a1) ? a1 : 0.f; o.z = isfinite(a2) ? a2 : 0.f; o.w = isfinite(a3) ? a3 : 0.f;
        *reinterpret_cast<float4*>(R + (size_t)idx * 4) = o;
    }
}

// ... and the rows that are NOT shared: the first pt / last pb output rows of each window see fewer filter rows.  Window b's edge
// row e (e < pt: y = e; else y = H - pb + e - pt) is written per window, already shifted so that the consumer's ordinary map
// rs * X + (bias - mean * rs * S[x][c]) gives the reference's value:  X = partial sum + mean_b * (S[x][c] - S_partial[x][c])
// (partial = the filter rows / columns that lie inside the window).  One thread per (b, e, x, 4 channels).
__global__ __launch_bounds__(256) void first_layer_edge_kernel(const float* __restrict__ mspec, const int32_t* __restrict__ win_row,
                                                               const float* __restrict__ stats, long long total, int H, int W, int Cout,
                                                               int kh, int kw, int pt, int pb, int pl, const float* __restrict__ w, int Kpad,
                                                               const float* __restrict__ S, float* __restrict__ E) {
    extern __shared__ __attribute__((aligned(16))) float sW[];     // [K][Cout]
    const int K = kh * kw;
    for (int e = threadIdx.x; e < K * Cout; e += 256) {
        const int co = e / K, k = e - co * K;
        sW[k * Cout + co] = w[(size_t)co * Kpad + k];
    }
    __syncthreads();
    const unsigned cg = (unsigned)Cout >> 2, utotal = (unsigned)total, ne = (unsigned)(pt + pb);
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < utotal; idx += gridDim.x * 256u) {
        const unsigned pix = idx / cg;
        const int c4 = (int)(idx - pix * cg) * 4;
        const unsigned be = pix / (unsigned)W;
        const int x = (int)(pix - be * (unsigned)W);
        const unsigned b = be / ne;
        const int e = (int)(be - b * ne);
        const int y = e < pt ? e : H - pb + (e - pt);
        const int wr = win_row[b];
        const float mean = stats[2u * b];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        for (int ky = 0; ky < kh; ++ky) {
            const int iy = y - pt + ky;
            if ((unsigned)iy >= (unsigned)H) continue;
            const float* src = mspec + (size_t)(wr + iy) * 24;
            for (int kx = 0; kx < kw; ++kx) {
                const int ix = x - pl + kx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float v = src[ix];
                const float4 wk = *reinterpret_cast<const float4*>(&sW[(ky * kw + kx) * Cout + c4]);
                a0 = fmaf(v, wk.x, a0); a1 = fmaf(v, wk.y, a1); a2 = fmaf(v, wk.z, a2); a3 = fmaf(v, wk.w, a3);
                s0 += wk.x; s1 += wk.y; s2 += wk.z; s3 += wk.w;
            }
        }
        const float4 sf = *reinterpret_cast<const float4*>(S + (size_t)x * Cout + c4);
        a0 = fmaf(mean, sf.x - s0, a0); a1 = fmaf(mean, sf.y - s1, a1); a2 = fmaf(mean, sf.z - s2, a2); a3 = fmaf(mean, sf.w - s3, a3);
        float4 o;
        o.x = isfinite(a0) ? a0 : 0.f; o.y = isfinite(a1) ? a1 : 0.f; o.z = isfinite(a2) ? a2 : 0.f; o.w = isfinite(a3) ? a3 : 0.f;
        *reinterpret_cast<float4*>(E + (size_t)idx * 4) = o;
    }
}

// The same R for a compile-time filter shape, one thread per (log-mel row t, 4 channels): the thread keeps the 24 values
// of each of its KH input rows in registers and produces ALL Wout positions of the row from them -- 6 float4
// loads per input row instead of KH * KW scalar loads per OUTPUT (first_layer_raw_kernel issues 20 loads per 16 bytes
// stored and ran at 1.8 TB/s of stores, bound by the load-issue rate of the texture path, not by HBM).  Every output is
// the same fmaf chain in (ky, kx) order: bit-identical results.  A store instruction covers 4 rows x 256 contiguous bytes.
template <int KH_, int KW_, int WOUT>
__global__ __launch_bounds__(256) void first_layer_rows_kernel(const float* __restrict__ mspec, int row0, int nrows, int Cout,
                                                               const float* __restrict__ w, int Kpad, float* __restrict__ R) {
    static_assert(WOUT + KW_ - 1 <= 24, "the log-mel rows are 24 wide");
    extern __shared__ __attribute__((aligned(16))) float sW[];     // [K][Cout]
    constexpr int K = KH_ * KW_;
    for (int e = threadIdx.x; e < K * Cout; e += 256) {
        const int co = e / K, k = e - co * K;
        sW[k * Cout + co] = w[(size_t)co * Kpad + k];
    }
    __syncthreads();
    const unsigned cg = (unsigned)Cout >> 2, total = (unsigned)nrows * cg;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned t = idx / cg;
        const int c4 = (int)(idx - t * cg) * 4;
        float4 acc[WOUT];
#pragma unroll
        for (int x = 0; x < WOUT; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < KH_; ++ky) {
            float v[24];
            const float4* src = reinterpret_cast<const float4*>(mspec + (size_t)(row0 + (int)t + ky) * 24);
#pragma unroll
            for (int q = 0; q < 6; ++q) { const float4 f = src[q]; v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w; }
#pragma unroll
            for (int kx = 0; kx < KW_; ++kx) {
                const float4 wk = *reinterpret_cast<const float4*>(&sW[(ky * KW_ + kx) * Cout + c4]);
#pragma unroll
                for (int x = 0; x < WOUT; ++x) {
                    acc[x].x = fmaf(v[x + kx], wk.x, acc[x].x);
                    acc[x].y = fmaf(v[x + kx], wk.y, acc[x].y);
                    acc[x].z = fmaf(v[x + kx], wk.z, acc[x].z);
                    acc[x].w = fmaf(v[x + kx], wk.w, acc[x].w);
                }
            }
        }
        float* dst = R + ((size_t)t * WOUT) * Cout + c4;
#pragma unroll
        for (int x = 0; x < WOUT; ++x) {
            float4 o;
            o.x = isfinite(acc[x].x) ? acc[x].x : 0.f; o.y = isfinite(acc[x].y) ? acc[x].y : 0.f;
            o.z = isfinite(acc[x].z) ? acc[x].z : 0.f; o.w = isfinite(acc[x].w) ? acc[x].w : 0.f;
            *reinterpret_cast<float4*>(dst + (size_t)x * Cout) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Direct 3x3 convolution of a ONE-channel image (stride 1, zero padding 1, bias, optional relu): ResNet-101's first layer
// (resnet.py:96-99, on the 64 x 144 fbank window of vbx_segmenter.py:262-266).  As a GEMM it has K = 9 padded to a 32-wide
// k-tile of scalar table-driven gathers (conv_x3_kernel<1>: 470 us per 512 windows, 4.5 x the time its 604 MB of output take to
// write).  Here a thread owns 8 consecutive rows x 4 channels of one column: 3 x 10 inputs in registers, 288 fmaf (f32, chain
// in (ky, kx) order), and a wave's store instruction covers 8 neighbouring columns x 128 B = 1 KB contiguous.
// Window input (vbx features, frame-major) and NHWC input share the addressing (row_stride / pix_stride / per-sample base).
template <bool WINDOW>
__global__ __launch_bounds__(256) void conv1_direct3x3_kernel(const ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sW[];     // [9][Cout]
    for (int e = threadIdx.x; e < 9 * p.Cout; e += 256) {
        const int co = e / 9, k = e - co * 9;
        sW[k * p.Cout + co] = p.w[(size_t)co * p.Kpad + k];
    }
    __syncthreads();
    const unsigned cg = (unsigned)p.Cout >> 2, hb = (unsigned)(p.H + 7) >> 3;
    const unsigned per_img = cg * (unsigned)p.W * hb;
    const unsigned total = per_img * (unsigned)(p.M / ((long long)p.H * p.W));       // host: < 2^32
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned b = idx / per_img;
        unsigned r = idx - b * per_img;
        const unsigned ob = r / (cg * (unsigned)p.W);
        r -= ob * cg * (unsigned)p.W;
        const int ox = (int)(r / cg), c4 = (int)(r - (unsigned)ox * cg) * 4, oy0 = (int)ob * 8;
        const float* src = p.in + (WINDOW ? (long long)p.win_row[b] * p.pix_stride : (long long)b * p.img_stride);
        float v[3][10];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox - 1 + dx;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int iy = oy0 - 1 + j;
                const bool ok = (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
                const float x = src[ok ? (long long)iy * p.row_stride + (long long)ix * p.pix_stride : 0];
                v[dx][j] = ok ? x : 0.f;
            }
        }
        float4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4 wk = *reinterpret_cast<const float4*>(&sW[(ky * 3 + kx) * p.Cout + c4]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i].x = fmaf(v[kx][i + ky], wk.x, acc[i].x);
                    acc[i].y = fmaf(v[kx][i + ky], wk.y, acc[i].y);
                    acc[i].z = fmaf(v[kx][i + ky], wk.z, acc[i].z);
                    acc[i].w = fmaf(v[kx][i + ky], wk.w, acc[i].w);
                }
            }
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + c4);
        float* dst = p.out + (((size_t)b * p.H + oy0) * p.W + ox) * p.Cout + c4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (oy0 + i >= p.H) break;
            float4 o = make_float4(acc[i].x + b4.x, acc[i].y + b4.y, acc[i].z + b4.z, acc[i].w + b4.w);
            if (p.act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            *reinterpret_cast<float4*>(dst + (size_t)i * p.W * p.Cout) = o;
        }
    }
}

// One wavefront per slot.  segmenter.py:82 (np.mean / np.std over the flattened window) and
// :86 (finite = all(isfinite(normalised))).
__global__ __launch_bounds__(256) void patch_stats_kernel(const float* __restrict__ mspec,
                                                          const int32_t* __restrict__ win_row, int n, int h,
                                                          float* __restrict__ stats, uint8_t* __restrict__ finite) {
    // A lane owns the flattened elements e = lane + 64 k of the 68 x h window (k < 23 for h <= 21, < 26 for h = 24): they are
    // loaded ONCE into registers and the three passes (mean, squared deviations, finite test) run on the registers, in
    // the element order and with the f64 sums of the three-pass form (bit-identical results).  (r, c) of an element
    // advance by (64 / h, 64 % h) with a carry: the run-time division per element and pass was most of this kernel's time.
    constexpr int NV = 26;                           // ceil(68 * 24 / 64)
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const float* src = mspec + (size_t)win_row[b] * 24;
    const int cnt = 68 * h;
    const int dr = 64 / h, dc = 64 - dr * h;
    int r = lane / h, c = lane - r * h;
    float v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int e = lane + 64 * k;
        v[k] = src[e < cnt ? r * 24 + c : 0];        // unconditional load (tail lanes re-read element 0 and are masked below)
        r += dr; c += dc;
        if (c >= h) { c -= h; ++r; }
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) if (lane + 64 * k < cnt) s += (double)v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean_d = s / cnt;
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < cnt) { const double d = (double)v[k] - mean_d; q += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float meanf = (float)mean_d;
    const float sdf = (float)sqrt(q / cnt);
    int bad = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < cnt) bad |= !isfinite((v[k] - meanf) / sdf);
    bad = __any(bad);
    if (lane == 0) { stats[2 * b] = meanf; stats[2 * b + 1] = sdf; finite[b] = bad ? 0 : 1; }
}

// NHWC pooling; one thread per (sample, oy, ox, c)
__global__ void pool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int H, int W,
                            int C, int Ho, int Wo, int kh, int kw, int sh, int sw, int pt, int pl, int kind) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const long long b = t / Ho;
    float acc = kind == 0 ? -INFINITY : 0.f;
    int cnt = 0;
    for (int ky = 0; ky < kh; ++ky) {
        const int iy = oy * sh - pt + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int ix = ox * sw - pl + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float v = in[((b * H + iy) * W + ix) * C + c];
            acc = kind == 0 ? fmaxf(acc, v) : acc + v;
            ++cnt;
        }
    }
    // average: over the window's elements INSIDE the input -- kh * kw unless the pool is zero-padded (Keras / TF 'same' average
    // pooling leaves the padding out of the mean)
    out[idx] = kind == 0 ? acc : acc / (float)(cnt > 0 ? cnt : 1);
}


// Elementwise activations that are not fused into a producer's epilogue (ISS_OP_ACT): elu, leaky relu, selu, softplus, clipped relu.
// One thread per 4 elements (the tail scalar), grid-stride; in place when in == out -- the op program's rows are, so the pointers are
// NOT __restrict__ (iss_cnn_load refuses an ISS_OP_ACT row that reads the network input).
__global__ __launch_bounds__(256) void act_kernel(const float* in, float* out, long long total, int act, float alpha, float p2, float p3) {
    auto f = [&](float v) {
        if (act == 9) return v > p3 ? fminf(v, p2) : alpha * (v - p3);                   // keras.layers.ReLU(max_value = p2, negative_slope = alpha, threshold = p3)
        if (act == 8) return fminf(fmaxf(v, 0.f), alpha);                                // keras.layers.ReLU(max_value = alpha)
        if (act == 4) return v > 0.f ? v : alpha * (expf(v) - 1.f);                       // keras.activations.elu
        if (act == 5) return v > 0.f ? v : alpha * v;                                    // keras.layers.LeakyReLU
        if (act == 6) return 1.05070098f * (v > 0.f ? v : 1.67326324f * (expf(v) - 1.f)); // selu
        return v > 20.f ? v : log1pf(expf(v));                                           // softplus
    };
    const long long n4 = total >> 2;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x = f(v.x); v.y = f(v.y); v.z = f(v.z); v.w = f(v.w);
        reinterpret_cast<float4*>(out)[i] = v;
    }
    for (long long i = (n4 << 2) + blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) out[i] = f(in[i]);
}

// Merge / data-movement rows of graph-shaped models (ISS_OP_ELT, include/iss.h): one thread per output element, grid-stride.  `in`,
// `res` and `out` may alias for the binary kinds (each element is read before it is written, by the same thread).
__global__ __launch_bounds__(256) void elt_kernel(const float* in, const float* res, float* out, long long total, int kind,
                                                  int cin, int cout, int nch, int soff, int doff, int d0, int d1, int d2,
                                                  int s0, int s1, int s2, int relu) {
    auto bin = [&](float a, float b) {
        const float v = kind == ISS_ELT_ADD ? a + b : kind == ISS_ELT_SUB ? a - b : kind == ISS_ELT_MUL ? a * b :
                        kind == ISS_ELT_MAX ? fmaxf(a, b) : kind == ISS_ELT_MIN ? fminf(a, b) : (a + b) * 0.5f;
        return relu ? fmaxf(v, 0.f) : v;
    };
    if (kind >= ISS_ELT_ADD && kind <= ISS_ELT_AVG && (total & 3) == 0) {      // float4 at a time (buffers are 256-byte aligned)
        const long long n4 = total >> 2;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 a = reinterpret_cast<const float4*>(in)[i], b = reinterpret_cast<const float4*>(res)[i];
            reinterpret_cast<float4*>(out)[i] = make_float4(bin(a.x, b.x), bin(a.y, b.y), bin(a.z, b.z), bin(a.w, b.w));
        }
        return;
    }
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        if (kind >= ISS_ELT_ADD && kind <= ISS_ELT_AVG) {
            out[i] = bin(in[i], res[i]);
        } else if (kind == ISS_ELT_COPY || kind == ISS_ELT_ZERO) {       // total = pixels * nch
            const long long p = i / nch;
            const int ch = (int)(i - p * nch);
            out[p * cout + doff + ch] = kind == ISS_ELT_ZERO ? 0.f : in[p * cin + soff + ch];
        } else {                                                         // PERMUTE: output (d0, d1, d2) row-major per sample, input strides s0..s2
            const long long per = (long long)d0 * d1 * d2;
            const long long smp = i / per;
            long long r = i - smp * per;
            const int i2 = (int)(r % d2); r /= d2;
            const int i1 = (int)(r % d1);
            const int i0 = (int)(r / d1);
            out[i] = in[smp * per + (long long)i0 * s0 + (long long)i1 * s1 + (long long)i2 * s2];
        }
    }
}

__global__ void softmax_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int C) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + r * C;
    float mx = x[0];
    for (int i = 1; i < C; ++i) mx = fmaxf(mx, x[i]);
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += expf(x[i] - mx);
    for (int i = 0; i < C; ++i) out[r * C + i] = expf(x[i] - mx) / s;
}

// mean || std over W (time) for every (h, c); out[(c*H + h)] and out[C*H + c*H + h]  (resnet.py:123-127)
__global__ void statpool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int H, int W, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int h = (int)(t % H);
    const long long b = t / H;
    const float* src = in + ((b * H + h) * (long long)W) * C + c;
    float s = 0.f, q = 0.f;
    for (int w = 0; w < W; ++w) { const float v = src[(long long)w * C]; s += v; q += v * v; }
    const float mean = s / W, meansq = q / W;
    float* o = out + b * (2LL * C * H);
    o[c * H + h] = mean;
    o[(long long)C * H + c * H + h] = sqrtf(meansq - mean * mean + 1e-10f);
}

__global__ void fill_half_kernel(float* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.5f;
}

// rows whose window was not finite get 0.5 everywhere (segmenter.py:175)
__global__ void mask_probs_kernel(float* probs, const uint8_t* finite, long long n, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * C && !finite[i / C]) probs[i] = 0.5f;
}

inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does); weights are finite
inline uint16_t bf16_rne(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// fused-pool window of a conv row (1,1 when absent)
inline void fused_pool_of(const int32_t* R, int& ph, int& pw) {
    ph = R[ISS_C_FPOOLH] > 1 ? R[ISS_C_FPOOLH] : 1;
    pw = R[ISS_C_FPOOLW] > 1 ? R[ISS_C_FPOOLW] : 1;
}

}  // namespace

// ============================================================================ host side
int iss_cnn_free(iss_ctx* c, int id) {
    if (!c || id < 0 || id >= ISS_MAX_NETS) return ISS_EINVAL;
    IssNet& n = c->nets[id];
    if (n.d_blob) (void)hipFree(n.d_blob);
    if (n.d_wh) (void)hipFree(n.d_wh);
    if (n.d_wl) (void)hipFree(n.d_wl);
    for (auto& kv : n.dhl_wp) if (kv.second) (void)hipFree(kv.second);
    if (n.d_wh16) (void)hipFree(n.d_wh16);
    if (n.d_wl16) (void)hipFree(n.d_wl16);
    if (n.d_wsum) (void)hipFree(n.d_wsum);
    if (n.d_ktab) (void)hipFree(n.d_ktab);
    n = IssNet();
    return ISS_OK;
}

extern "C" int iss_cnn_load(iss_ctx* c, int id, const int32_t* prog, int32_t nrows, const float* blob,
                            int64_t blob_floats, int32_t nbuf, const int64_t* buf_elems, int32_t in_h, int32_t in_w,
                            int32_t in_c, int32_t out_dim) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || !prog || nrows <= 0 || !blob || blob_floats <= 0 || nbuf <= 0 || !buf_elems)
        return iss_fail(c, ISS_EINVAL, "iss_cnn_load: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    iss_cnn_free(c, id);
    IssNet& n = c->nets[id];
    n.prog.assign(prog, prog + (size_t)nrows * ISS_PROG_COLS);
    n.nrows = nrows; n.nbuf = nbuf; n.buf_elems.assign(buf_elems, buf_elems + nbuf);
    n.in_h = in_h; n.in_w = in_w; n.in_c = in_c; n.out_dim = out_dim;
    n.kpad.assign(nrows, 0); n.ktab_off.assign(nrows, -1);
    std::vector<int32_t> ktab;
    double flops = 0;
    for (int r = 0; r < nrows; ++r) {
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        auto bad = [&](const char* what) { return iss_fail(c, ISS_EINVAL, "iss_cnn_load: row %d: %s", r, what); };
        if (R[ISS_C_IN] != ISS_BUF_INPUT && (R[ISS_C_IN] < 0 || R[ISS_C_IN] >= nbuf)) return bad("IN buffer id");
        if (R[ISS_C_OUT] < 0 || R[ISS_C_OUT] >= nbuf) return bad("OUT buffer id");
        if (R[ISS_C_OP] == ISS_OP_CONV) {
            const int K = R[ISS_C_KH] * R[ISS_C_KW] * R[ISS_C_CIN];
            const int Kpad = roundup(K, KALIGN);
            if (K <= 0 || R[ISS_C_COUT] <= 0) return bad("conv shape");
            if (R[ISS_C_WOFF] < 0 || (R[ISS_C_WOFF] & 7) || (int64_t)R[ISS_C_WOFF] + (int64_t)R[ISS_C_COUT] * Kpad > blob_floats)
                return bad("weight offset misaligned or outside blob (weights must be [Cout][roundup32(K)], offset % 8 == 0)");
            if (R[ISS_C_RES] >= nbuf) return bad("RES buffer id");
            if (R[ISS_C_KH] > 32767 || R[ISS_C_KW] > 32767) return bad("kernel too large");
            int fph, fpw;
            fused_pool_of(R, fph, fpw);
            if (fph * fpw != 1 && fph * fpw != 2 && fph * fpw != 4) return bad("fused pool window must cover 2 or 4 outputs");
            if (fph * fpw > 1 && (R[ISS_C_RES] >= 0 || R[ISS_C_HO] / fph < 1 || R[ISS_C_WO] / fpw < 1))
                return bad("fused pool with residual / empty pooled output");
            const bool patch = R[ISS_C_INMODE] == 1;
            const bool window = R[ISS_C_INMODE] == 2;      // (H = features, W = frames) view of the resident (T, H) vbx features
            if (patch && (R[ISS_C_CIN] != 1 || R[ISS_C_H] != 68 || R[ISS_C_W] > 24)) return bad("patch-mode conv must read (68, <=24, 1)");
            if (window && (R[ISS_C_CIN] != 1 || R[ISS_C_H] != 64)) return bad("window-mode conv must read (64, frames, 1)");
            const int rs = patch ? 24 : (window ? 1 : R[ISS_C_W] * R[ISS_C_CIN]);
            const int ps = patch ? 1 : (window ? R[ISS_C_H] : R[ISS_C_CIN]);
            n.kpad[r] = Kpad; n.ktab_off[r] = (int64_t)ktab.size();
            for (int k = 0; k < Kpad; ++k) {
                if (k < K) {
                    const int cc = k % R[ISS_C_CIN], kk = k / R[ISS_C_CIN];
                    const int kx = kk % R[ISS_C_KW], ky = kk / R[ISS_C_KW];
                    ktab.push_back(ky * rs + kx * ps + cc);
                    ktab.push_back((ky << 16) | kx);
                } else {                              // K padding: force the bounds test to fail -> zeros
                    ktab.push_back(0);
                    ktab.push_back((0x7fff << 16) | 0x7fff);
                }
            }
            flops += 2.0 * K * R[ISS_C_COUT] * (double)(R[ISS_C_HO] / fph * fph) * (double)(R[ISS_C_WO] / fpw * fpw);
            if (R[ISS_C_DUALW] != 0 || R[ISS_C_DUALB] != 0) {     // projection shortcut (row r - 1) + expansion (row r) as one GEMM
                if (r < 1) return bad("ISS_C_DUALW on the first row");
                const int32_t* P = &n.prog[(size_t)(r - 1) * ISS_PROG_COLS];
                int pph, ppw;
                fused_pool_of(P, pph, ppw);
                const bool shapes = P[ISS_C_OP] == ISS_OP_CONV && P[ISS_C_KH] == 1 && P[ISS_C_KW] == 1 && R[ISS_C_KH] == 1 && R[ISS_C_KW] == 1 &&
                                    R[ISS_C_SH] == 1 && R[ISS_C_SW] == 1 && P[ISS_C_SH] >= 1 && P[ISS_C_SW] >= 1 &&
                                    R[ISS_C_PT] == 0 && R[ISS_C_PL] == 0 && P[ISS_C_PT] == 0 && P[ISS_C_PL] == 0 &&
                                    R[ISS_C_INMODE] == 0 && P[ISS_C_INMODE] == 0 && fph * fpw == 1 && pph * ppw == 1 &&
                                    R[ISS_C_HO] == R[ISS_C_H] && R[ISS_C_WO] == R[ISS_C_W] && P[ISS_C_HO] == R[ISS_C_HO] &&
                                    P[ISS_C_WO] == R[ISS_C_WO] && P[ISS_C_COUT] == R[ISS_C_COUT] &&
                                    (P[ISS_C_HO] - 1) * P[ISS_C_SH] < P[ISS_C_H] && (P[ISS_C_WO] - 1) * P[ISS_C_SW] < P[ISS_C_W] &&
                                    R[ISS_C_CIN] % 32 == 0 && P[ISS_C_CIN] % 32 == 0;
                const bool chain = R[ISS_C_RES] >= 0 && R[ISS_C_RES] == P[ISS_C_OUT] && R[ISS_C_OUT] == R[ISS_C_RES] &&
                                   R[ISS_C_IN] != P[ISS_C_OUT] && P[ISS_C_IN] != P[ISS_C_OUT] && P[ISS_C_RES] < 0 && P[ISS_C_ACT] == 0 &&
                                   P[ISS_C_PSOFF] < 0 && R[ISS_C_PSOFF] < 0 && P[ISS_C_BOFF] >= 0 && R[ISS_C_BOFF] >= 0;
                const int64_t wo = (int64_t)R[ISS_C_DUALW] - 1, bo = (int64_t)R[ISS_C_DUALB] - 1;
                const bool offs = wo >= 0 && bo >= 0 && (wo & 7) == 0 &&
                                  wo + (int64_t)R[ISS_C_COUT] * (R[ISS_C_CIN] + P[ISS_C_CIN]) <= blob_floats && bo + R[ISS_C_COUT] <= blob_floats;
                if (!shapes || !chain || !offs)
                    return bad("ISS_C_DUALW / ISS_C_DUALB: rows r - 1, r are not a linear 1x1 projection and the in-place 1x1 expansion it is added to, "
                               "or the concatenated parameters lie outside the blob (include/iss.h)");
            }
        } else if (R[ISS_C_OP] == ISS_OP_ACT) {
            if (R[ISS_C_ACT] < 4 || R[ISS_C_ACT] > 9) return bad("ISS_OP_ACT: activation code must be 4 (elu), 5 (leaky relu), 6 (selu), 7 (softplus), 8 (relu with max_value) or 9 (keras ReLU in full)");
            if (R[ISS_C_IN] == ISS_BUF_INPUT) return bad("ISS_OP_ACT: an elementwise activation cannot read the network input (it works in place)");
        } else if (R[ISS_C_OP] == ISS_OP_ELT) {
            const int k = R[ISS_C_ACT];
            const long long hw = (long long)R[ISS_C_H] * R[ISS_C_W];
            if (k < ISS_ELT_COPY || k > ISS_ELT_PERMUTE) return bad("ISS_OP_ELT: unknown kind");
            if (R[ISS_C_HO] < 1 || R[ISS_C_WO] < 1 || R[ISS_C_COUT] < 1 || R[ISS_C_CIN] < 1 || hw < 1) return bad("ISS_OP_ELT: shape");
            if ((long long)R[ISS_C_HO] * R[ISS_C_WO] * R[ISS_C_COUT] > buf_elems[R[ISS_C_OUT]]) return bad("ISS_OP_ELT: output larger than its buffer");
            if (R[ISS_C_IN] != ISS_BUF_INPUT && k != ISS_ELT_ZERO && hw * R[ISS_C_CIN] > buf_elems[R[ISS_C_IN]]) return bad("ISS_OP_ELT: input larger than its buffer");
            if (R[ISS_C_IN] == ISS_BUF_INPUT && k != ISS_ELT_ZERO) return bad("ISS_OP_ELT: a merge row cannot read the network input");
            if (k >= ISS_ELT_ADD && k <= ISS_ELT_AVG) {
                if (R[ISS_C_RES] < 0 || R[ISS_C_RES] >= nbuf) return bad("ISS_OP_ELT: a binary row needs its second operand in ISS_C_RES");
                if (R[ISS_C_HO] != R[ISS_C_H] || R[ISS_C_WO] != R[ISS_C_W] || R[ISS_C_COUT] != R[ISS_C_CIN]) return bad("ISS_OP_ELT: binary rows keep the shape");
                if (hw * R[ISS_C_CIN] > buf_elems[R[ISS_C_RES]]) return bad("ISS_OP_ELT: second operand larger than its buffer");
            } else if (k == ISS_ELT_COPY || k == ISS_ELT_ZERO) {
                if (R[ISS_C_HO] != R[ISS_C_H] || R[ISS_C_WO] != R[ISS_C_W]) return bad("ISS_OP_ELT: COPY / ZERO keep the pixel grid");
                if (R[ISS_C_KH] < 1 || R[ISS_C_PL] < 0 || R[ISS_C_PL] + R[ISS_C_KH] > R[ISS_C_COUT]) return bad("ISS_OP_ELT: destination channel range outside COUT");
                if (k == ISS_ELT_COPY && (R[ISS_C_PT] < 0 || R[ISS_C_PT] + R[ISS_C_KH] > R[ISS_C_CIN] || R[ISS_C_IN] == R[ISS_C_OUT]))
                    return bad("ISS_OP_ELT: COPY source channel range outside CIN, or IN == OUT");
            } else {
                const int pm[3] = {R[ISS_C_KH], R[ISS_C_KW], R[ISS_C_SH]};
                const int dims[3] = {R[ISS_C_H], R[ISS_C_W], R[ISS_C_CIN]};
                if (pm[0] < 0 || pm[0] > 2 || pm[1] < 0 || pm[1] > 2 || pm[2] < 0 || pm[2] > 2 || pm[0] == pm[1] || pm[0] == pm[2] || pm[1] == pm[2])
                    return bad("ISS_OP_ELT: PERMUTE needs a permutation of (0, 1, 2) in KH, KW, SH");
                if (R[ISS_C_HO] != dims[pm[0]] || R[ISS_C_WO] != dims[pm[1]] || R[ISS_C_COUT] != dims[pm[2]] || R[ISS_C_IN] == R[ISS_C_OUT])
                    return bad("ISS_OP_ELT: PERMUTE output shape is not the permuted input shape, or IN == OUT");
            }
        } else if (R[ISS_C_OP] != ISS_OP_POOL && R[ISS_C_OP] != ISS_OP_SOFTMAX && R[ISS_C_OP] != ISS_OP_STATPOOL) {
            return bad("unknown op");
        }
    }
    n.flops_per_sample = flops; n.blob_floats = blob_floats;
    ISS_HIP(c, hipMalloc((void**)&n.d_blob, (size_t)blob_floats * sizeof(float)));
    ISS_HIP(c, hipMemcpy(n.d_blob, blob, (size_t)blob_floats * sizeof(float), hipMemcpyHostToDevice));
    {   // bf16 hi / lo parts of every parameter, same offsets as the f32 blob (conv_x3_kernel operands)
        std::vector<uint16_t> hi((size_t)blob_floats), lo((size_t)blob_floats);
        for (int64_t i = 0; i < blob_floats; ++i) {
            hi[i] = bf16_rne(blob[i]);
            lo[i] = bf16_rne(blob[i] - bf16_to_f32(hi[i]));
        }
        // fp16 split of the same values (ISS_PREC_F16X3); a value outside fp16's range makes the whole network ineligible
        std::vector<uint16_t> hi16((size_t)blob_floats), lo16((size_t)blob_floats);
        n.f16_ok = true;
        for (int64_t i = 0; i < blob_floats; ++i) {
            const float x = blob[i];
            if (!(std::fabs(x) < 65504.f)) { n.f16_ok = false; hi16[i] = lo16[i] = 0; continue; }
            const _Float16 h = (_Float16)x;
            const _Float16 l = (_Float16)(x - (float)h);
            memcpy(&hi16[i], &h, 2); memcpy(&lo16[i], &l, 2);
        }
        ISS_HIP(c, hipMalloc((void**)&n.d_wh16, (size_t)blob_floats * 2 + 16));
        ISS_HIP(c, hipMalloc((void**)&n.d_wl16, (size_t)blob_floats * 2 + 16));
        ISS_HIP(c, hipMemcpy(n.d_wh16, hi16.data(), (size_t)blob_floats * 2, hipMemcpyHostToDevice));
        ISS_HIP(c, hipMemcpy(n.d_wl16, lo16.data(), (size_t)blob_floats * 2, hipMemcpyHostToDevice));
        ISS_HIP(c, hipMalloc((void**)&n.d_wh, (size_t)blob_floats * 2 + 16));
        ISS_HIP(c, hipMalloc((void**)&n.d_wl, (size_t)blob_floats * 2 + 16));
        ISS_HIP(c, hipMemcpy(n.d_wh, hi.data(), (size_t)blob_floats * 2, hipMemcpyHostToDevice));
        ISS_HIP(c, hipMemcpy(n.d_wl, lo.data(), (size_t)blob_floats * 2, hipMemcpyHostToDevice));
    }
    {   // per-channel weight sums of the patch-mode first layers (ConvArgs::f_wsum)
        std::vector<float> wsum;
        n.wsum_off.assign(nrows, -1);
        n.wsumx_off.assign(nrows, -1);
        for (int r = 0; r < nrows; ++r) {
            const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
            if (R[ISS_C_OP] != ISS_OP_CONV || R[ISS_C_INMODE] != 1) continue;
            const int K = R[ISS_C_KH] * R[ISS_C_KW] * R[ISS_C_CIN];
            n.wsum_off[r] = (int64_t)wsum.size();
            for (int co = 0; co < R[ISS_C_COUT]; ++co) {
                double acc = 0.0;
                for (int k = 0; k < K; ++k) acc += (double)blob[R[ISS_C_WOFF] + (int64_t)co * n.kpad[r] + k];
                wsum.push_back((float)acc);
            }
            while (wsum.size() % 8) wsum.push_back(0.f);         // float4-aligned rows
            // zero-padded first layer: S[x][co] = sum over all filter rows and the filter columns that see data at column x
            // (conv_x3_ws_kernel<..., FS>, first_layer_edge_kernel)
            if ((R[ISS_C_PT] != 0 || R[ISS_C_PL] != 0 || R[ISS_C_WO] == R[ISS_C_W]) && R[ISS_C_CIN] == 1 && R[ISS_C_SW] == 1) {
                n.wsumx_off[r] = (int64_t)wsum.size();
                const int W = R[ISS_C_W], kh = R[ISS_C_KH], kw = R[ISS_C_KW], pl = R[ISS_C_PL];
                for (int x = 0; x < W; ++x)
                    for (int co = 0; co < R[ISS_C_COUT]; ++co) {
                        double acc = 0.0;
                        for (int ky = 0; ky < kh; ++ky)
                            for (int kx = 0; kx < kw; ++kx)
                                if (x - pl + kx >= 0 && x - pl + kx < W) acc += (double)blob[R[ISS_C_WOFF] + (int64_t)co * n.kpad[r] + ky * kw + kx];
                        wsum.push_back((float)acc);
                    }
                while (wsum.size() % 8) wsum.push_back(0.f);
            }
        }
        if (!wsum.empty()) {
            ISS_HIP(c, hipMalloc((void**)&n.d_wsum, wsum.size() * sizeof(float)));
            ISS_HIP(c, hipMemcpy(n.d_wsum, wsum.data(), wsum.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (!ktab.empty()) {
        ISS_HIP(c, hipMalloc((void**)&n.d_ktab, ktab.size() * sizeof(int32_t)));
        ISS_HIP(c, hipMemcpy(n.d_ktab, ktab.data(), ktab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    n.loaded = true;
    return ISS_OK;
}

extern "C" int iss_set_precision(iss_ctx* c, int mode) {
    if (!c) return ISS_EINVAL;
    if (mode != ISS_PREC_BF16X3 && mode != ISS_PREC_F32 && mode != ISS_PREC_F16X3) return iss_fail(c, ISS_EINVAL, "iss_set_precision: unknown mode %d", mode);
    c->precision = mode;
    return ISS_OK;
}

extern "C" int iss_set_precision_guard(iss_ctx* c, float threshold) {
    if (!c) return ISS_EINVAL;
    if (!(threshold == threshold)) return iss_fail(c, ISS_EINVAL, "iss_set_precision_guard: threshold is NaN");
    c->guard_threshold = threshold;
    return ISS_OK;
}

extern "C" int iss_cnn_set_net_precision(iss_ctx* c, int id, int mode) {
    if (!c || id < 0 || id >= ISS_MAX_NETS) return ISS_EINVAL;
    if (mode != -1 && mode != ISS_PREC_BF16X3 && mode != ISS_PREC_F32 && mode != ISS_PREC_F16X3) return iss_fail(c, ISS_EINVAL, "iss_cnn_set_net_precision: unknown mode %d", mode);
    if (!c->nets[id].loaded) return iss_fail(c, ISS_ESTATE, "net %d not loaded", id);
    c->nets[id].prec_override = mode;
    c->nets[id].guard_state = mode == -1 ? ISS_GUARD_PENDING : ISS_GUARD_FIXED;
    return ISS_OK;
}

extern "C" int iss_cnn_precision_info(iss_ctx* c, int id, int32_t* mode, float* max_dlogp, int32_t* slots, int32_t* state, float* dlogp_in_use) {
    if (!c || id < 0 || id >= ISS_MAX_NETS) return ISS_EINVAL;
    const IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "net %d not loaded", id);
    if (mode) *mode = n.prec_override >= 0 ? n.prec_override : c->precision;
    if (max_dlogp) *max_dlogp = n.guard_dlogp;
    if (slots) *slots = n.guard_slots;
    if (state) *state = n.guard_state;
    if (dlogp_in_use) *dlogp_in_use = n.guard_dlogp_chosen;
    return ISS_OK;
}

extern "C" int iss_set_diag(iss_ctx* c, uint32_t flags) {
    if (!c) return ISS_EINVAL;
    if (flags & ~(uint32_t)ISS_DIAG_ALL) return iss_fail(c, ISS_EINVAL, "iss_set_diag: unknown bits 0x%x", flags & ~(uint32_t)ISS_DIAG_ALL);
    c->diag = flags;
    return ISS_OK;
}

extern "C" int iss_cnn_flops(iss_ctx* c, int id, double* f) {
    if (!c || id < 0 || id >= ISS_MAX_NETS || !f) return ISS_EINVAL;
    if (!c->nets[id].loaded) return iss_fail(c, ISS_ESTATE, "net %d not loaded", id);
    *f = c->nets[id].flops_per_sample;
    return ISS_OK;
}

namespace {

// Kernel shapes conv_x3_fp_kernel is instantiated for (the tap loop is unrolled at compile time);
// other shapes run on conv_x3_kernel.
// conv_x3_ws_kernel decomposes a flattened window pixel p < limit as p / W == (p * ceil(2^16 / W)) >> 16: exact iff
// limit * (ceil(2^16 / W) * W - 2^16) < 2^16
inline bool ws_recip_exact(int W, long long limit) {
    const long long m = (65536 + W - 1) / W;
    return limit * (m * W - 65536) < 65536 && limit * m < (1ll << 31);
}
inline bool ws_shape_compiled(int kh, int kw) {
#define ISS_WS_HAS(KH_, KW_) if (kh == KH_ && kw == KW_) return true;
    ISS_WS_SHAPES(ISS_WS_HAS)
#undef ISS_WS_HAS
    return false;
}
inline bool fp_shape_compiled(int kh, int kw) {
#define ISS_FP_HAS(KH_, KW_) if (kh == KH_ && kw == KW_) return true;
    ISS_FP_SHAPES(ISS_FP_HAS)
#undef ISS_FP_HAS
    return false;
}

// Host replica of the device's row mapping / footprint arithmetic: does every 128-row tile of this
// launch touch at most FPIX pixels?  The pattern is periodic in the sample index (period <= BM
// samples), so tiles covering the first BM + 2 samples decide.
int footprint_pixels(const ConvArgs& a, int TM = BM) {      // largest pixel span of a TM-row tile of this launch (INT_MAX: irregular)
    const long long rows_per_sample = (long long)a.Hq * a.Wq * a.pp;
    const long long samples = a.M / rows_per_sample;
    const long long lim_rows = std::min<long long>(a.M, rows_per_sample * std::min<long long>(samples, TM + 2));
    auto pix_of = [&](long long m, int ky, int kx) {
        long long q = m;
        int dy = 0, dx = 0;
        if (a.pp > 1) { q = m / a.pp; const int j = (int)(m - q * a.pp); dy = j / a.pw; dx = j - dy * a.pw; }
        const int hw = a.Hq * a.Wq;
        const long long b = q / hw;
        const int rem = (int)(q - b * hw);
        const int qy = rem / a.Wq, qx = rem - qy * a.Wq;
        const int oy = qy * a.ph + dy, ox = qx * a.pw + dx;
        return (b * a.H + (oy * a.sh - a.pt_ + ky)) * a.W + (ox * a.sw - a.pl_ + kx);
    };
    long long worst = 0;
    for (long long m0 = 0; m0 < lim_rows; m0 += TM) {
        const long long m_last = std::min<long long>(m0 + TM, a.M) - 1;
        const long long lo = pix_of(m0, 0, 0), hi = pix_of(m_last, a.H_k - 1, a.kw - 1);
        worst = std::max<long long>(worst, hi - lo + 1);
        // rows inside the tile never reach below lo / above hi (row-major or pool-window-major order); check anyway
        for (long long m = m0; m <= m_last; ++m)
            if (pix_of(m, 0, 0) < lo || pix_of(m, a.H_k - 1, a.kw - 1) > hi) return 0x7fffffff;
    }
    return (int)std::min<long long>(worst, 0x7fffffff);
}

// Run the op program on `bc` samples.  src: PATCH mode uses (d_winrow + s0, stats, finite),
// otherwise `d_input` is an NHWC batch.  The result is left in act[last OUT].  [rmin, rmax] = range of the window rows
// of this call; share_first: the caller's (chunking-independent) decision to use the shared first layer.
int run_program(iss_ctx* c, IssNet& n, int bc, const int32_t* d_winrow, const float* d_stats,
                const uint8_t* d_fin, const float* d_input, float** result, int rmin = 0, int rmax = -1,
                bool share_first = false) {
    const int prec = n.prec_override >= 0 ? n.prec_override : c->precision;          // (precision guard: one network may run exact f32)
    const bool x3mode = prec != ISS_PREC_F32;                    // a split-operand mode (bf16 or fp16 halves)
    // ISS_PREC_F16X3: the launches with an fp16 instantiation (the one-wave-per-SIMD conv2 / conv3 / conv4 kernels and the long-K
    // dense kernel: > 99.9 % of the segmenter nets' arithmetic) take fp16 operand halves; a SMALL layer without one runs in exact
    // f32 (conv_igemm_kernel: tests/precision_emulation.py -- the last dense layers in bf16 halves would undo most of the gain),
    // anything else keeps bf16 halves
    const bool f16mode = prec == ISS_PREC_F16X3 && n.f16_ok && n.d_wh16 != nullptr;
    double net_flops = 0.0;
    if (f16mode)
        for (int q = 0; q < n.nrows; ++q) {
            const int32_t* Q = &n.prog[(size_t)q * ISS_PROG_COLS];
            if (Q[ISS_C_OP] == ISS_OP_CONV) net_flops += 2.0 * Q[ISS_C_KH] * Q[ISS_C_KW] * Q[ISS_C_CIN] * (double)Q[ISS_C_COUT] * Q[ISS_C_HO] * Q[ISS_C_WO];
        }
    // A PATCH first layer directly in front of a footprint-kernel conv is not launched per window: it is computed once
    // per log-mel row and the second conv normalises it per window while staging its LDS footprint (ConvArgs::f_*,
    // conv_fp.h FUSED).  Static part of the test; the footprint-capacity part is decided when the second row is reached
    // (the first layer is then launched per window after all).
    auto can_defer = [&](int r) {
        if (r + 1 >= n.nrows || !share_first || rmax < rmin) return false;
        // exact-f32 mode: only the shape the F32 form of the weight-stationary kernel is instantiated for (conv_ws.h F32, cnn_ws_h.hip)
        const bool f32defer = !x3mode;
        if (f32defer && (c->diag & (ISS_DIAG_NO_WS | ISS_DIAG_NO_F32WS))) return false;
        const int32_t* R1 = &n.prog[(size_t)r * ISS_PROG_COLS];
        const int32_t* R2 = &n.prog[(size_t)(r + 1) * ISS_PROG_COLS];
        int ph, pw;
        fused_pool_of(R1, ph, pw);
        const bool pool1 = ph * pw != 1;                 // fused non-overlapping pool behind the first layer: max only, gather path only
        if (R1[ISS_C_OP] != ISS_OP_CONV || R1[ISS_C_INMODE] != 1 || R1[ISS_C_RES] >= 0 || R1[ISS_C_ACT] > 1) return false;
        if (pool1 && (R1[ISS_C_POOLKIND] != 0 || !x3mode || (c->diag & ISS_DIAG_NO_GFUSED))) return false;
        if (R1[ISS_C_CIN] != 1 || R1[ISS_C_SH] != 1 || R1[ISS_C_SW] != 1) return false;
        const bool valid1 = R1[ISS_C_PT] == 0 && R1[ISS_C_PL] == 0 && R1[ISS_C_HO] == R1[ISS_C_H] - R1[ISS_C_KH] + 1 &&
                            R1[ISS_C_WO] == R1[ISS_C_W] - R1[ISS_C_KW] + 1;                                                     // 'valid'
        // 'same' (zero-padded, output = input size): shared through conv_x3_ws_kernel<..., FS> (S table + per-window edge rows)
        const bool same_geo = !valid1 && R1[ISS_C_HO] == R1[ISS_C_H] && R1[ISS_C_WO] == R1[ISS_C_W] && R1[ISS_C_PT] <= R1[ISS_C_KH] - 1 &&
                              R1[ISS_C_PL] <= R1[ISS_C_KW] - 1 && R1[ISS_C_KH] <= R1[ISS_C_H] && n.wsumx_off[r] >= 0 && R1[ISS_C_PSOFF] < 0 &&
                              !(c->diag & ISS_DIAG_NO_FSAME);
        const bool same_ws = same_geo && R1[ISS_C_W] * R1[ISS_C_COUT] * 4 <= issk::WS_STAB && !(c->diag & ISS_DIAG_NO_WS) &&
                             issk::iss_ws_fs_compiled(R2[ISS_C_KH], R2[ISS_C_KW]);
        // ... or through the generic gather kernel (conv_x3_kernel<4>: any second conv)
        const bool same1 = same_ws || (same_geo && x3mode && !(c->diag & ISS_DIAG_NO_GFUSED));
        if (!valid1 && !same1) return false;
        if (pool1 && !valid1) return false;
        if (f32defer && (!valid1 || !issk::iss_ws_f32_fused_compiled(R2[ISS_C_KH], R2[ISS_C_KW]) || R2[ISS_C_SH] != 1 || R2[ISS_C_SW] != 1 ||
                         R2[ISS_C_PT] != 0 || R2[ISS_C_PL] != 0 || R1[ISS_C_PSOFF] >= 0)) return false;
        if (R1[ISS_C_KH] * R1[ISS_C_KW] * R1[ISS_C_COUT] * 4 > 48 * 1024) return false;        // first_layer_raw_kernel's LDS weights
        if (R1[ISS_C_BOFF] < 0 || (R1[ISS_C_PSOFF] >= 0) != (R1[ISS_C_PTOFF] >= 0) || R1[ISS_C_COUT] % 4 != 0 || n.wsum_off[r] < 0) return false;
        if (R2[ISS_C_OP] != ISS_OP_CONV || R2[ISS_C_INMODE] != 0 || R2[ISS_C_IN] != R1[ISS_C_OUT] || R2[ISS_C_RES] >= 0) return false;
        if (R2[ISS_C_CIN] != R1[ISS_C_COUT] || R2[ISS_C_CIN] % XBK != 0 || R2[ISS_C_H] != R1[ISS_C_HO] / ph || R2[ISS_C_W] != R1[ISS_C_WO] / pw) return false;
        const bool ring2 = valid1 && issk::iss_ws_ring_compiled(R2[ISS_C_KH], R2[ISS_C_KW]) && !(c->diag & (ISS_DIAG_NO_RING | ISS_DIAG_NO_WS));
        // a footprint kernel can take it: (a zero-padded second conv is fused by the weight-stationary kernel only; conv_row decides);
        // the footprint may touch two windows at most, and the x / W trick of the kernel needs a small W
        const bool foot2 = R2[ISS_C_KH] * R2[ISS_C_KW] >= 8 && (fp_shape_compiled(R2[ISS_C_KH], R2[ISS_C_KW]) || ring2) &&   // (>= 12 unless the weight-stationary kernel takes it, see conv_row)
                           R2[ISS_C_H] * R2[ISS_C_W] >= FPIX + 32 && R2[ISS_C_W] <= 128 && (valid1 || same_ws) && !pool1;
        // ... or the generic gather kernel reads the shared rows itself (conv_x3_kernel<3>): any second conv, 'valid' first layer
        const bool gath2 = x3mode && (valid1 || same_geo) && R1[ISS_C_PSOFF] < 0 && !(c->diag & ISS_DIAG_NO_GFUSED);
        if (!foot2 && !gath2) return false;
        for (int q = r + 2; q < n.nrows; ++q) {                  // nobody else may read the first layer's output
            const int32_t* Q = &n.prog[(size_t)q * ISS_PROG_COLS];
            if (Q[ISS_C_IN] == R1[ISS_C_OUT] || Q[ISS_C_RES] == R1[ISS_C_OUT]) return false;
            if (Q[ISS_C_OUT] == R1[ISS_C_OUT]) break;
        }
        return true;
    };
    // rows r, r + 1: an in-place 1x1 stride-1 expansion with identity residual and relu, then a plain 1x1 stride-1 convolution to
    // 32 / 64 / 128 channels that reads it (the next Bottleneck's reduction, resnet.py:48-58) -- the pair conv_x3_pwc_kernel computes
    auto chain_pair = [&](int r) {
        if (r + 1 >= n.nrows) return false;
        const int32_t* R1 = &n.prog[(size_t)r * ISS_PROG_COLS];
        const int32_t* R2 = &n.prog[(size_t)(r + 1) * ISS_PROG_COLS];
        int ph, pw, ph2, pw2;
        fused_pool_of(R1, ph, pw);
        fused_pool_of(R2, ph2, pw2);
        auto plain1x1 = [](const int32_t* R) {
            return R[ISS_C_OP] == ISS_OP_CONV && R[ISS_C_KH] == 1 && R[ISS_C_KW] == 1 && R[ISS_C_SH] == 1 && R[ISS_C_SW] == 1 &&
                   R[ISS_C_PT] == 0 && R[ISS_C_PL] == 0 && R[ISS_C_INMODE] == 0 && R[ISS_C_PSOFF] < 0 && R[ISS_C_BOFF] >= 0 &&
                   R[ISS_C_HO] == R[ISS_C_H] && R[ISS_C_WO] == R[ISS_C_W];
        };
        return plain1x1(R1) && plain1x1(R2) && ph * pw == 1 && ph2 * pw2 == 1 && R1[ISS_C_DUALW] == 0 &&
               R1[ISS_C_RES] >= 0 && R1[ISS_C_RES] == R1[ISS_C_OUT] && R1[ISS_C_IN] != R1[ISS_C_OUT] && R1[ISS_C_IN] != ISS_BUF_INPUT &&
               R1[ISS_C_ACT] == 1 && R2[ISS_C_IN] == R1[ISS_C_OUT] && R2[ISS_C_RES] < 0 && R2[ISS_C_OUT] != R1[ISS_C_OUT] &&
               R2[ISS_C_OUT] != R1[ISS_C_IN] && R2[ISS_C_ACT] <= 1 && R2[ISS_C_CIN] == R1[ISS_C_COUT] && issk::pwc_compiled(R1[ISS_C_CIN], R2[ISS_C_COUT]) &&
               R2[ISS_C_H] == R1[ISS_C_HO] && R2[ISS_C_W] == R1[ISS_C_WO] && n.kpad[r] == R1[ISS_C_CIN] && n.kpad[r + 1] == R2[ISS_C_CIN];
    };
#ifdef ISS_PW_NO_ASM_RING                            // build-time escape (Makefile): none of the asm-load kernels of conv_pw.h / conv_pwc.h
    constexpr bool asm_ring_ok = false;
#else
    constexpr bool asm_ring_ok = true;
#endif
    // ---- CHL hand-over between footprint kernels (conv_common.h, round 6).  hl_np[buffer] = pixels per plane while the tensor in
    // that activation buffer is in the CHL layout (absent: f32 NHWC).  A producer writes CHL only when the NEXT row is the tensor's
    // only reader and runs on conv_x3_wq3h_kernel (wq3_plan: the conditions conv_row launches conv_x3_wq3_kernel under).
    std::map<int, unsigned> hl_np;
    std::map<int, bool> hl_f16;                                  // ... and holds fp16 (not bf16) planes
    std::map<int, bool> hl_dense;                                // ... and is the flattened-feature CHL tensor of a dense layer (conv_dhl.h)
    bool hl_out_dense = false;
    bool hl_out_f16 = false;
    int hl_out_row = -1;                                         // the row conv_row has just launched with a CHL output ...
    unsigned hl_out_np = 0;                                      // ... and its plane size
    const bool no_hl = (c->diag & ISS_DIAG_NO_HL) != 0;
    auto wq3_plan = [&](int q, int* tmr_out) -> int {            // -1, or the kind (0: bias + relu, 1: relu + 2 x 1 max-pool)
        if (q < 0 || q >= n.nrows || !x3mode) return -1;
        if (c->diag & (ISS_DIAG_NO_WS | ISS_DIAG_NO_WS3 | ISS_DIAG_NO_WQ)) return -1;
        const int32_t* R = &n.prog[(size_t)q * ISS_PROG_COLS];
        if (R[ISS_C_OP] != ISS_OP_CONV || R[ISS_C_INMODE] != 0 || R[ISS_C_IN] == ISS_BUF_INPUT || R[ISS_C_RES] >= 0 || R[ISS_C_DUALW] != 0) return -1;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.H = R[ISS_C_H]; a.W = R[ISS_C_W]; a.Cin = R[ISS_C_CIN]; a.Cout = R[ISS_C_COUT];
        fused_pool_of(R, a.ph, a.pw);
        a.pp = a.ph * a.pw;
        a.poolkind = R[ISS_C_POOLKIND];
        a.Hq = R[ISS_C_HO] / a.ph; a.Wq = R[ISS_C_WO] / a.pw;
        a.H_k = R[ISS_C_KH]; a.kw = R[ISS_C_KW];
        a.sh = R[ISS_C_SH]; a.sw = R[ISS_C_SW]; a.pt_ = R[ISS_C_PT]; a.pl_ = R[ISS_C_PL];
        a.act = R[ISS_C_ACT];
        a.M = (long long)bc * a.Hq * a.Wq * a.pp;
        a.img_stride = (long long)a.H * a.W * a.Cin;
        const bool padded = a.pt_ != 0 || a.pl_ != 0 || (R[ISS_C_HO] - 1) * a.sh - a.pt_ + a.H_k > a.H || (R[ISS_C_WO] - 1) * a.sw - a.pl_ + a.kw > a.W;
        if (a.Cin % XBK != 0 || padded || a.sh != 1 || a.sw != 1 || a.Cout % (2 * BN) != 0 || !issk::iss_ws_nh2_compiled(a.H_k, a.kw) || a.H_k != 3 || a.kw != 3 ||
            a.Cin % F2_CH != 0 || a.Cin < 2 * F2_CH || a.M >= (1ll << 31) || (long long)bc * a.img_stride * 4 >= (1ll << 32)) return -1;
        if (R[ISS_C_BOFF] < 0 || a.act != 1 || R[ISS_C_PSOFF] >= 0 || !ws_recip_exact(a.W, issk::WQ3_PIX + a.W)) return -1;
        {
            const long long key = ((long long)q << 32) | (unsigned)bc | (1ll << 60);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            if (it->second > WS_PIX2) return -1;
        }
        int kind = -1;
        if (a.pp == 1 && a.M * (long long)a.Cout * 4 < 0xFFF00000ll) kind = 0;
        else if (a.pp == 2 && a.ph == 2 && a.poolkind == 0 && (a.M / 2) * (long long)a.Cout * 4 < 0xFFF00000ll) kind = 1;
        if (kind < 0) return -1;
        const long long key = ((long long)q << 32) | (unsigned)bc | (1ll << 58);
        auto it = n.fp_pix.find(key);
        if (it == n.fp_pix.end()) {
            int tmr = 0;
            for (int cand = issk::WQ3_TM; cand >= issk::WQ3_TM - 64 && !tmr; cand -= 4)
                if (footprint_pixels(a, cand) <= issk::WQ3_PIX) tmr = cand;
            it = n.fp_pix.emplace(key, tmr).first;
        }
        if (it->second <= 0) return -1;
        if (tmr_out) *tmr_out = it->second;
        return kind;
    };
    // row r's output (Cout channels, `npix` pixels for this call) may be written in the CHL layout: row r + 1 is a conv_x3_wq3h_kernel
    // launch that reads it, and nobody else does before the buffer is written again
    auto want_hl_out = [&](int r, long long npix) -> bool {
        if (no_hl || r + 1 >= n.nrows) return false;
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        const int32_t* Q = &n.prog[(size_t)(r + 1) * ISS_PROG_COLS];
        const int ob = R[ISS_C_OUT];
        if (Q[ISS_C_IN] != ob || Q[ISS_C_OUT] == ob || Q[ISS_C_CIN] != R[ISS_C_COUT] || R[ISS_C_COUT] % (2 * BN) != 0 && R[ISS_C_COUT] != BN) return false;
        if (npix != (long long)bc * Q[ISS_C_H] * Q[ISS_C_W] || !issk::chl_fits(npix, R[ISS_C_COUT])) return false;
        if (wq3_plan(r + 1, nullptr) < 0) return false;
        for (int t = r + 2; t < n.nrows; ++t) {
            const int32_t* T = &n.prog[(size_t)t * ISS_PROG_COLS];
            if (T[ISS_C_IN] == ob || T[ISS_C_RES] == ob) return false;
            if (T[ISS_C_OUT] == ob) break;
        }
        return true;
    };
    // row r is a pooled conv launch (conv_x3_wq3h_kernel<1, ..>) whose output, flattened, is read by the dense layer of row r + 1 and by
    // nobody else: it may write the CHL tensor conv_dhl_kernel fetches by LDS-DMA (window = "pixel", feature = "channel")
    auto want_dhl_out = [&](int r, int hq, int wq) -> bool {
        if (no_hl || r + 1 >= n.nrows || !x3mode) return false;
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        const int32_t* Q = &n.prog[(size_t)(r + 1) * ISS_PROG_COLS];
        const int ob = R[ISS_C_OUT];
        const long long K = (long long)hq * wq * R[ISS_C_COUT];
        int qph, qpw;
        fused_pool_of(Q, qph, qpw);
        if (Q[ISS_C_OP] != ISS_OP_CONV || Q[ISS_C_IN] != ob || Q[ISS_C_OUT] == ob || Q[ISS_C_INMODE] != 0 || Q[ISS_C_RES] >= 0 || Q[ISS_C_DUALW] != 0) return false;
        if (Q[ISS_C_KH] != 1 || Q[ISS_C_KW] != 1 || Q[ISS_C_H] != 1 || Q[ISS_C_W] != 1 || Q[ISS_C_HO] != 1 || Q[ISS_C_WO] != 1 || qph * qpw != 1) return false;
        if (Q[ISS_C_CIN] != K || n.kpad[r + 1] != K || hq * wq < 2 || R[ISS_C_COUT] % 8 != 0) return false;
        if (!issk::dhl_supported((int)K, Q[ISS_C_COUT], Q[ISS_C_ACT], Q[ISS_C_PSOFF] >= 0, false)) return false;
        const size_t bytes = (size_t)issk::dhl_npad(bc) * (size_t)K * 4;
        if (bytes > (size_t)bc * K * 4 + issk::ISS_ACT_SLACK || bytes >= 0xFFF00000ull || (long long)bc * hq * wq * (hq * wq) >= (1ll << 32)) return false;
        if ((long long)bc * Q[ISS_C_COUT] * 4 >= (1ll << 32)) return false;
        for (int t = r + 2; t < n.nrows; ++t) {
            const int32_t* T = &n.prog[(size_t)t * ISS_PROG_COLS];
            if (T[ISS_C_IN] == ob || T[ISS_C_RES] == ob) return false;
            if (T[ISS_C_OUT] == ob) break;
        }
        return true;
    };
    constexpr int kDualDeclined = -12345;                        // conv_row(r, -1, r - 1): the two-source launch is not possible for this call
    std::function<int(int, int, int, int)> conv_row = [&](int r, int pend, int dual, int chain) -> int {
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        const float* in = R[ISS_C_IN] == ISS_BUF_INPUT ? d_input : (const float*)c->act[R[ISS_C_IN]].p;
        float* out = (float*)c->act[R[ISS_C_OUT]].p;
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.in = in;
        a.w = n.d_blob + R[ISS_C_WOFF];
        a.bias = R[ISS_C_BOFF] >= 0 ? n.d_blob + R[ISS_C_BOFF] : nullptr;
        a.ps = R[ISS_C_PSOFF] >= 0 ? n.d_blob + R[ISS_C_PSOFF] : nullptr;
        a.pt = R[ISS_C_PTOFF] >= 0 ? n.d_blob + R[ISS_C_PTOFF] : nullptr;
        a.res = R[ISS_C_RES] >= 0 ? (const float*)c->act[R[ISS_C_RES]].p : nullptr;
        a.out = out;
        a.ktab = n.d_ktab + n.ktab_off[r];
        a.wh = n.d_wh + R[ISS_C_WOFF];
        a.wl = n.d_wl + R[ISS_C_WOFF];
        a.H = R[ISS_C_H]; a.W = R[ISS_C_W]; a.Cin = R[ISS_C_CIN]; a.Cout = R[ISS_C_COUT];
        fused_pool_of(R, a.ph, a.pw);
        a.pp = a.ph * a.pw;
        a.poolkind = R[ISS_C_POOLKIND];
        a.Hq = R[ISS_C_HO] / a.ph; a.Wq = R[ISS_C_WO] / a.pw;
        a.H_k = R[ISS_C_KH]; a.kw = R[ISS_C_KW];
        a.sh = R[ISS_C_SH]; a.sw = R[ISS_C_SW]; a.pt_ = R[ISS_C_PT]; a.pl_ = R[ISS_C_PL];
        a.act = R[ISS_C_ACT]; a.Kpad = n.kpad[r];
        a.M = (long long)bc * a.Hq * a.Wq * a.pp;
        const bool patch = R[ISS_C_INMODE] == 1;
        // (fp16 mode) a small layer -- under 0.5 % of the network's arithmetic -- that no fp16 kernel takes: exact f32
        const double row_flops = 2.0 * R[ISS_C_KH] * R[ISS_C_KW] * R[ISS_C_CIN] * (double)R[ISS_C_COUT] * R[ISS_C_HO] * R[ISS_C_WO];
        const bool small_cand = f16mode && pend < 0 && row_flops < 0.005 * net_flops && row_flops < 2e6 &&     // (and small in absolute terms: a
                                R[ISS_C_INMODE] == 0 && !can_defer(r);                                        //  ResNet-101 has 105 layers under 1 %)
        // ... unless it is a dense layer of some width (a 512 -> 512 head: 0.5 MFLOP per window, 66 TFLOP/s on conv_igemm_kernel, 5 % of
        // such a net's step): conv_x3_pw_kernel has an fp16 form for any K, so it takes the layer instead of the streaming kernels
        const bool f16_dense_pw = small_cand && row_flops >= 2.5e5 && a.H_k == 1 && a.kw == 1 && a.H == 1 && a.W == 1 && R[ISS_C_HO] == 1 &&
                                  R[ISS_C_WO] == 1 && a.pp == 1 && a.sh == 1 && a.sw == 1 && a.pt_ == 0 && a.pl_ == 0 && a.Cout % 4 == 0 && a.Cin % XBK == 0 &&
                                  a.Kpad == a.Cin && !a.res &&
                                  (c->diag & ISS_DIAG_NO_PW) == 0;
        const bool small_row = small_cand && !f16_dense_pw;
        const bool x3 = x3mode && !small_row;
        bool row_f16 = f16mode;                                  // cleared below where the launch has no fp16 form
        const bool in_is_hl = R[ISS_C_IN] != ISS_BUF_INPUT && hl_np.count(R[ISS_C_IN]) != 0;     // (only conv_x3_wq3h_kernel reads that layout)
        bool in_hl_taken = false;
        if (in_is_hl && hl_dense.count(R[ISS_C_IN])) {
            // the first dense layer on the flattened-feature CHL tensor conv4 wrote for it (want_dhl_out): both operands by LDS-DMA
            const bool f16 = hl_f16.count(R[ISS_C_IN]) != 0;
            const int K = R[ISS_C_CIN];
            uint16_t*& wp = n.dhl_wp[{r, f16 ? 1 : 0}];
            if (!wp) {
                ISS_HIP(c, hipMalloc((void**)&wp, issk::dhl_packed_elems(K, R[ISS_C_COUT]) * 2));
                issk::iss_dhl_pack((f16 ? n.d_wh16 : n.d_wh) + R[ISS_C_WOFF], (f16 ? n.d_wl16 : n.d_wl) + R[ISS_C_WOFF], wp, R[ISS_C_COUT], n.kpad[r], K, c->stream);
            }
            issk::DhlArgs d;
            d.a = reinterpret_cast<const uint16_t*>(in); d.wp = wp; d.bias = a.bias; d.out = out;
            d.np = hl_np[R[ISS_C_IN]]; d.M = bc; d.K = K; d.Cout = R[ISS_C_COUT]; d.act = R[ISS_C_ACT];
            iss_prof_begin(c, 0, 2.0 * K * (double)R[ISS_C_COUT] * (double)bc);
            iss_prof_tag(c, ISS_PROF_PW);
            iss_prof_row(c, r);
            iss_prof_inst(c, "conv_dhl_kernel<%s,%d>", f16 ? "true" : "false", ISS_DHL_NW);       // <F16,NW>
            issk::iss_dhl_launch(d, c->stream, f16);
            iss_prof_end(c);
            return ISS_OK;
        }
        a.mode = patch ? 2 : ((a.Cin % (x3 ? XBK : 4) == 0) ? 0 : 1);
        const bool window = R[ISS_C_INMODE] == 2;
        if (patch) {
            if (!d_winrow) return iss_fail(c, ISS_ESTATE, "patch-mode network run without a window list");
            a.in = (const float*)c->mspec.p; a.win_row = d_winrow; a.stats = d_stats; a.finite = d_fin;
            a.row_stride = 24; a.pix_stride = 1; a.img_stride = 0;
        } else if (window) {
            if (!d_winrow || !c->vbx_out.p) return iss_fail(c, ISS_ESTATE, "window-mode network run without resident vbx features");
            a.in = (const float*)c->vbx_out.p; a.win_row = d_winrow;
            a.row_stride = 1; a.pix_stride = a.H; a.img_stride = 0;
            a.mode = 1;
        } else {
            if (!in) return iss_fail(c, ISS_ESTATE, "network input missing");
            a.row_stride = a.W * a.Cin; a.pix_stride = a.Cin; a.img_stride = (long long)a.H * a.W * a.Cin;
        }
        a.nblk = (unsigned)((a.M + BM - 1) / BM);
        set_fast_div(a, 0, a.pp); set_fast_div(a, 1, a.pw); set_fast_div(a, 2, a.Hq * a.Wq); set_fast_div(a, 3, a.Wq);
#ifdef ISS_EXPERIMENTS
        { static const int dbg = getenv("ISS_DBG") ? atoi(getenv("ISS_DBG")) : 0; a.dbg = dbg; }     // timing-only experiment bits (never in a release build)
#endif
        a.nblk_n = (unsigned)((a.Cout + BN - 1) / BN);
        dim3 grid(a.nblk, a.nblk_n);
        const dim3 grid1(a.nblk * a.nblk_n);          // generic kernels: 1-D, XCD-aware (gemm_tile_of_block)
        double fl = 2.0 * R[ISS_C_KH] * R[ISS_C_KW] * a.Cin * (double)a.Cout * (double)a.M;
        if (chain >= 0) {
            // row r (in-place 1x1 expansion + identity residual + relu) and row `chain` = r + 1 (the next Bottleneck's 1x1 reduction
            // to 128 channels, which reads row r's output) as ONE launch: the reduction consumes x' out of LDS (conv_pwc.h)
            const int32_t* Q = &n.prog[(size_t)chain * ISS_PROG_COLS];
            a.wh2 = n.d_wh + Q[ISS_C_WOFF]; a.wl2 = n.d_wl + Q[ISS_C_WOFF];
            a.bias2 = Q[ISS_C_BOFF] >= 0 ? n.d_blob + Q[ISS_C_BOFF] : nullptr;
            a.out2 = (float*)c->act[Q[ISS_C_OUT]].p;
            a.act2 = Q[ISS_C_ACT]; a.Cout2 = Q[ISS_C_COUT];
            if (!x3 || a.mode != 0 || !in || !issk::pwc_supported(a)) return kDualDeclined;
            fl += 2.0 * Q[ISS_C_CIN] * (double)Q[ISS_C_COUT] * (double)a.M;
            iss_prof_begin(c, 0, fl);
            iss_prof_tag(c, ISS_PROF_PW);
            iss_prof_row(c, r);
            iss_prof_inst(c, "conv_x3_pwc_kernel<%d,%d>", a.Cin / 32, a.Cout2 / 32);
            issk::iss_pwc_launch(a, c->stream);
            iss_prof_end(c);
            return ISS_OK;
        }
        if (dual >= 0) {
            // rows `dual` (a linear 1x1 projection, any stride) and r (the in-place 1x1 expansion it is added to) as ONE GEMM over
            // both inputs on the concatenated weights (ISS_C_DUALW; validated by iss_cnn_load): row `dual`'s output never exists
            const int32_t* P = &n.prog[(size_t)dual * ISS_PROG_COLS];
            a.in2 = P[ISS_C_IN] == ISS_BUF_INPUT ? d_input : (const float*)c->act[P[ISS_C_IN]].p;
            a.Cin2 = P[ISS_C_CIN]; a.H2 = P[ISS_C_H]; a.W2 = P[ISS_C_W]; a.sh2 = P[ISS_C_SH]; a.sw2 = P[ISS_C_SW];
            const int64_t wo = (int64_t)R[ISS_C_DUALW] - 1;
            a.w = n.d_blob + wo; a.wh = n.d_wh + wo; a.wl = n.d_wl + wo;
            a.bias = n.d_blob + (R[ISS_C_DUALB] - 1);
            a.res = nullptr;
            a.Kpad = a.Cin + a.Cin2;
            if (!x3 || a.mode != 0 || !in || !a.in2 || !issk::pws2_dual_supported(a)) return kDualDeclined;
            fl += 2.0 * a.Cin2 * (double)a.Cout * (double)a.M;
            iss_prof_begin(c, 0, fl);
            iss_prof_tag(c, ISS_PROF_PW);
            iss_prof_row(c, r);
            iss_prof_inst(c, "conv_x3_pws2_kernel<true,false,dual>");
            issk::iss_pws2_launch(a, c->stream, false, true);
            iss_prof_end(c);
            return ISS_OK;
        }
        bool fp = false;                               // LDS-footprint kernel usable
        if (x3 && a.mode == 0 && fp_shape_compiled(a.H_k, a.kw) && a.M < (1ll << 31)) {
            const long long key = ((long long)r << 32) | (unsigned)bc;
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a)).first;
            fp = it->second <= FPIX;
        }
        // weight-stationary kernel (conv_ws.h): the shared-first-layer convolution, 8..16 taps, one N tile of 64 channels
        bool ws = false;
        const bool no_ws = (c->diag & ISS_DIAG_NO_WS) != 0;
        // the deferred first layer in front is zero-padded ('same'): only the FS form of the weight-stationary kernel can fuse it
        const bool fs1 = pend >= 0 && (n.prog[(size_t)pend * ISS_PROG_COLS + ISS_C_HO] == n.prog[(size_t)pend * ISS_PROG_COLS + ISS_C_H]);
        int ph1 = 1, pw1 = 1;                            // the deferred first layer's own fused (max) pool: the gather form only
        if (pend >= 0) fused_pool_of(&n.prog[(size_t)pend * ISS_PROG_COLS], ph1, pw1);
        const bool pool1 = ph1 * pw1 != 1;
        if (!no_ws && fp && pend >= 0 && a.H_k * a.kw >= 8 && a.H_k * a.kw <= WS_MAXNT && ws_shape_compiled(a.H_k, a.kw) &&
            a.Cin % F2_CH == 0 && a.H * a.W >= WS_PIX + 64 + (a.pt_ + 1) * a.W && ws_recip_exact(a.W, a.H * a.W + WS_PIX + a.W)) {
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 62);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            ws = it->second <= WS_PIX && n.prog[(size_t)pend * ISS_PROG_COLS + ISS_C_PSOFF] < 0;     // (a post-activation affine of the
        }                                                                                            //  first layer stays on conv_x3_fp_kernel)
        // ring form (conv_ws.h RING): more than WS_MAXNT taps (7x7), first-layer-fused, one 512-row tile per group on a
        // 1024-pixel footprint; row-major epilogue
        bool ws_ring = false;
        if (!no_ws && !(c->diag & ISS_DIAG_NO_RING) && pend >= 0 && !fs1 && x3 && a.mode == 0 && issk::iss_ws_ring_compiled(a.H_k, a.kw) &&
            a.sh == 1 && a.sw == 1 && a.Cin % F2_CH == 0 && a.Cin >= 2 * F2_CH && a.M < (1ll << 31) &&
            ws_recip_exact(a.W, a.H * a.W + WS_PIX2 + a.W) && n.prog[(size_t)pend * ISS_PROG_COLS + ISS_C_PSOFF] < 0) {
            // rows per tile: the largest multiple of 4 (<= 512, >= 320) whose footprint fits the 1024 pixels -- a 512-row tile of a
            // pooled 59 x 14 output under a 7-row filter spans 1036 pixels, 496 rows 1002 (ConvArgs::tmr; the rest of the tile idles)
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 57);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) {
                // ... and reaches into at most ONE following window (the fetch decomposes a footprint position into two windows)
                const int cap = std::min<long long>(WS_PIX2, (long long)a.H * a.W - 64 - (long long)(a.pt_ + 1) * a.W);
                int tmr = 0;
                for (int cand = WS_TM; cand >= 320 && !tmr; cand -= 4)
                    if (footprint_pixels(a, cand) <= cap) tmr = cand;
                it = n.fp_pix.emplace(key, tmr).first;
            }
            a.tmr = it->second;
            ws_ring = a.tmr > 0 && a.M % 4 == 0;
            if (ws_ring) { ws = true; fp = true; } else a.tmr = 0;
        }
        const bool padded = a.pt_ != 0 || a.pl_ != 0 ||
                            (R[ISS_C_HO] - 1) * a.sh - a.pt_ + a.H_k > a.H || (R[ISS_C_WO] - 1) * a.sw - a.pl_ + a.kw > a.W;
        // exact-f32 mode (ISS_PREC_F32): the F32 form of the weight-stationary kernel for the first-layer-fused 5x3 layer
        const bool no_f32ws = (c->diag & ISS_DIAG_NO_F32WS) != 0;
        bool ws_f32 = false;
        if (!x3 && !no_ws && !no_f32ws && pend >= 0 && !fs1 && a.mode == 0 && issk::iss_ws_f32_fused_compiled(a.H_k, a.kw) && !padded &&
            a.sh == 1 && a.sw == 1 && issk::epi_is_pool_relu(a) && a.Cin % F2_CH == 0 && a.Cin >= 2 * F2_CH && a.M < (1ll << 31) &&
            a.H * a.W >= WS_PIX + 64 + a.W && ws_recip_exact(a.W, a.H * a.W + WS_PIX + a.W) &&
            n.prog[(size_t)pend * ISS_PROG_COLS + ISS_C_PSOFF] < 0) {
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 62);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            ws_f32 = it->second <= WS_PIX;
            if (ws_f32) { ws = true; fp = true; }
        }
        // FS form: row-major epilogue only.  A second conv WITHOUT a fused pool is not taken: its unpooled output through the row-major
        // epilogue (4-byte stores) measured 3.1 -> 2.3 h/s on conv1_same_nopool against the per-window first layer + transposed kernel
        // ... except the one transposed instantiation: unpadded 5x3 with bias + relu (cnn_ws_f.hip)
        const bool fs_tr = a.pp == 1 && a.Cout % 4 == 0;
        const bool fs_tr_ok = fs_tr && a.H_k == 5 && a.kw == 3 && !padded && issk::epi_is_simple_tr(a);
        const bool ws_fs = fs1 && ws && issk::iss_ws_fs_compiled(a.H_k, a.kw) && (!fs_tr || fs_tr_ok) && a.sh == 1 && a.sw == 1 && a.W * a.Cin * 4 <= issk::WS_STAB &&
                           a.Cin >= 2 * F2_CH && !(c->diag & ISS_DIAG_NO_FSAME);
        if (fs1 && !ws_fs) ws = false;
        // weight-stationary kernel with two column halves per workgroup (conv_ws.h, NH = 2): unpadded 3x3 stride-1 layers with
        // a multiple of 128 output channels whose 512-row tiles fit a 1024-pixel footprint -- the 3x3 layers of the segmenter nets
        bool ws_nh2 = false;
        const bool no_ws3 = (c->diag & ISS_DIAG_NO_WS3) != 0;
        const bool nh2_pad_pool = a.pp > 1 && issk::epi_is_pool_relu(a);                         // ... and row-major with the pooled relu epilogue
        const bool nh2_pad_ok = (a.pp == 1 && a.Cout % 4 == 0 && issk::epi_is_simple_tr(a)) || nh2_pad_pool;   // the padded form is compiled transposed + simple
        // (exact-f32 mode: the unpadded form with the simple transposed or the pooled relu epilogue only -- cnn_ws_h.hip)
        const bool nh2_f32 = !x3 && !no_f32ws && !padded && ((a.pp == 1 && a.Cout % 4 == 0 && issk::epi_is_simple_tr(a)) || (a.pp > 1 && issk::epi_is_pool_relu(a)));
        if (!no_ws && !no_ws3 && pend < 0 && (x3 || nh2_f32) && a.mode == 0 && (!padded || nh2_pad_ok) && a.sh == 1 && a.sw == 1 && !a.res && a.Cout % (2 * BN) == 0 &&
            issk::iss_ws_nh2_compiled(a.H_k, a.kw) && a.Cin % F2_CH == 0 && a.M < (1ll << 31) &&
            (long long)bc * a.img_stride * 4 < (1ll << 32)) {
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 60);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            ws_nh2 = it->second <= WS_PIX2;
        }
        // plain weight-stationary launch: a padded 3x3 stride-1 layer too wide for the 360-pixel footprint kernel (see conv_ws.h)
        bool ws_plain = false;
        if (!no_ws && !fp && pend < 0 && x3 && a.mode == 0 && padded && a.sh == 1 && a.sw == 1 && a.pp == 1 && a.Cout % 4 == 0 &&
            !a.res && issk::iss_ws_plain_compiled(a.H_k, a.kw) && a.Cin % F2_CH == 0 && a.M < (1ll << 31) &&
            (long long)bc * a.img_stride * 4 < (1ll << 32)) {
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 61);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            ws_plain = it->second <= WS_PIX;
        }
        // ... and the UNPADDED 3x3 stride-1 layers the two-column-half form does not take (64 / 96 output channels): they ran on
        // conv_x3_fp_kernel (weights streamed per tap, 215-312 TF); simple transposed epilogue or pooled relu only (cnn_ws_c.hip)
        bool ws_plain_u = false;
        if (!no_ws && !(c->diag & ISS_DIAG_NO_WSU3) && !ws_nh2 && pend < 0 && x3 && a.mode == 0 && !padded && a.sh == 1 && a.sw == 1 && !a.res &&
            issk::iss_ws_plain_compiled(a.H_k, a.kw) && a.Cin % F2_CH == 0 && a.Cin >= 2 * F2_CH && a.M < (1ll << 31) &&
            (long long)bc * a.img_stride * 4 < (1ll << 32) &&
            ((a.pp == 1 && a.Cout % 4 == 0 && issk::epi_is_simple_tr(a)) || (a.pp > 1 && issk::epi_is_pool_relu(a)))) {
            const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 56);
            auto it = n.fp_pix.find(key);
            if (it == n.fp_pix.end()) it = n.fp_pix.emplace(key, footprint_pixels(a, WS_TM)).first;
            ws_plain_u = it->second <= WS_PIX;
        }
        bool fused = false;
        if (pend >= 0) {
            const int32_t* Rp = &n.prog[(size_t)pend * ISS_PROG_COLS];
            const long long edge_rows = fs1 ? (long long)bc * (Rp[ISS_C_KH] - 1) : 0;                          // per-window edge rows behind R
            fused = !pool1 && fp && (ws || (!fs1 && !padded && a.H_k * a.kw >= 12)) && d_winrow != nullptr &&
                    ((long long)(rmax - rmin) + Rp[ISS_C_HO] + edge_rows) * Rp[ISS_C_WO] * Rp[ISS_C_COUT] * 4 < (1ll << 32);   // 32-bit BYTE offsets into R
            if (ws_ring && !fused) { ws = false; fp = false; ws_ring = false; a.tmr = 0; }                                // (no footprint kernel of that shape)
            if (ws_f32 && !fused) { ws = false; fp = false; ws_f32 = false; }                                              // (exact-f32 mode has no other one)
        }
        if (!fused && (long long)bc * a.img_stride >= (1ll << 32)) fp = false;        // 32-bit offsets into the input batch
        // no footprint kernel fuses it and none would run this conv anyway: the generic gather kernel reads the shared first-layer
        // rows itself and applies the window's affine map + activation before its operand split (conv_x3_kernel<3>) -- the per-window
        // first-layer tensor (283-333 KB per slot) is neither written nor read for ANY second conv on overlapping windows
        bool gfused = false;
        if (pend >= 0 && !fused && x3 && a.mode == 0 && d_winrow != nullptr && !(c->diag & ISS_DIAG_NO_GFUSED) && !(fs1 && (c->diag & ISS_DIAG_NO_FSAME))) {
            const int32_t* Rp = &n.prog[(size_t)pend * ISS_PROG_COLS];
            gfused = (fs1 || (Rp[ISS_C_PT] == 0 && Rp[ISS_C_PL] == 0)) && Rp[ISS_C_PSOFF] < 0 && Rp[ISS_C_ACT] <= 1 && a.M < (1ll << 31) &&
                     (!fs1 || n.wsumx_off[pend] >= 0);
            if (gfused && fp) {
                // conv_x3_fp_kernel would run this conv (unfused) at ~330 TFLOP/s where the gather kernel does ~230, but needs the
                // per-window first-layer tensor, written at ~2.1 TB/s (measured: conv1_patch_x3_kernel): the gather kernel wins when
                // flops * (1/230e12 - 1/330e12) < bytes / 2.1e12, i.e. below ~360 flops per byte of that tensor (narrow nets)
                const double bytes1 = (double)bc * Rp[ISS_C_HO] * Rp[ISS_C_WO] * Rp[ISS_C_COUT] * 4.0;
                gfused = fl < 360.0 * bytes1;
            }
            if (gfused) fp = false;
        }
        if (pend >= 0) {
            if (!fused && !gfused) {                         // the deferred first layer runs on its own after all
                const int rc = conv_row(pend, -1, -1, -1);
                if (rc) return rc;
            } else {
                const int32_t* R1 = &n.prog[(size_t)pend * ISS_PROG_COLS];
                const long long rrows = (long long)(rmax - rmin) + R1[ISS_C_HO];     // R: rows rmin .. rmax + H1 - 1
                const long long rtot = rrows * R1[ISS_C_WO] * (R1[ISS_C_COUT] / 4);
                const int f_ne = fs1 ? R1[ISS_C_KH] - 1 : 0;                         // zero-padded first layer: edge rows per window
                const long long etot = (long long)bc * f_ne * R1[ISS_C_WO] * (R1[ISS_C_COUT] / 4);
                // pooled planes (ph1 of them) + the transformed window list behind R (see pool_rows_kernel)
                const int plane_rows = pool1 ? (int)(rrows / ph1) + 2 : 0;
                const long long ptot = pool1 ? (long long)ph1 * plane_rows * (R1[ISS_C_WO] / pw1) * (R1[ISS_C_COUT] / 4) : 0;
                { const int rc = iss_reserve(c, c->raw1, (size_t)(rtot + etot + ptot) * 16 + (pool1 ? (size_t)bc * 4 + 16 : 0)); if (rc) return rc; }
                float* Rraw = (float*)c->raw1.p;
                iss_prof_begin(c, 2, 0);
                if (fs1) {
                    // shared rows (all filter rows, the filter columns that see data) + the per-window edge rows behind them
                    const int pt1 = R1[ISS_C_PT], pl1 = R1[ISS_C_PL], pb1 = R1[ISS_C_KH] - 1 - pt1;
                    const float* S = n.d_wsum + n.wsumx_off[pend];
                    const size_t lds1 = (size_t)R1[ISS_C_KH] * R1[ISS_C_KW] * R1[ISS_C_COUT] * 4;
                    if (rtot >= (1ll << 31) || etot >= (1ll << 31)) return iss_fail(c, ISS_EINVAL, "internal: shared first layer over %lld + %lld items", rtot, etot);
                    hipLaunchKernelGGL(first_layer_same_kernel, dim3((unsigned)std::min<long long>((rtot + 255) / 256, 4096)), dim3(256), lds1, c->stream,
                                       (const float*)c->mspec.p, (int)c->T, rmin, rtot, R1[ISS_C_WO], R1[ISS_C_COUT], R1[ISS_C_KH], R1[ISS_C_KW],
                                       pt1, pl1, (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], Rraw);
                    if (etot > 0)
                        hipLaunchKernelGGL(first_layer_edge_kernel, dim3((unsigned)std::min<long long>((etot + 255) / 256, 8192)), dim3(256), lds1, c->stream,
                                           (const float*)c->mspec.p, d_winrow, d_stats, etot, R1[ISS_C_H], R1[ISS_C_WO], R1[ISS_C_COUT], R1[ISS_C_KH],
                                           R1[ISS_C_KW], pt1, pb1, pl1, (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], S, Rraw + (size_t)rtot * 4);
                    a.f_padt = pt1; a.f_padb = pb1; a.f_erow0 = (int)rrows;
                } else {
                const dim3 rgrid((unsigned)std::min<long long>((rtot + 255) / 256, 4096));
                const size_t rlds = (size_t)R1[ISS_C_KH] * R1[ISS_C_KW] * R1[ISS_C_COUT] * 4;
                const bool no_rows = (c->diag & ISS_DIAG_NO_FLROWS) != 0;             // diagnostic: the per-output kernel
                // compiled for the two input widths of the reference's nets: 21 bands (smn / sm) -> 17 positions, 24 (gender) -> 20
                const bool rows_ok = !no_rows && R1[ISS_C_KH] == 4 && R1[ISS_C_KW] == 5 && (R1[ISS_C_WO] == 17 || R1[ISS_C_WO] == 20) &&
                                     rrows * (R1[ISS_C_COUT] / 4) < (1ll << 31);
                const dim3 rowgrid((unsigned)std::min<long long>((rrows * (R1[ISS_C_COUT] / 4) + 255) / 256, 4096));
                if (rows_ok && R1[ISS_C_WO] == 17)
                    hipLaunchKernelGGL((first_layer_rows_kernel<4, 5, 17>), rowgrid, dim3(256), rlds, c->stream, (const float*)c->mspec.p, rmin, (int)rrows,
                                       R1[ISS_C_COUT], (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], Rraw);
                else if (rows_ok)
                    hipLaunchKernelGGL((first_layer_rows_kernel<4, 5, 20>), rowgrid, dim3(256), rlds, c->stream, (const float*)c->mspec.p, rmin, (int)rrows,
                                       R1[ISS_C_COUT], (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], Rraw);
                else if (R1[ISS_C_KH] == 4 && R1[ISS_C_KW] == 5)
                    hipLaunchKernelGGL((first_layer_raw_kernel<4, 5>), rgrid, dim3(256), rlds, c->stream, (const float*)c->mspec.p, rmin, rtot,
                                       R1[ISS_C_WO], R1[ISS_C_COUT], 4, 5, (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], Rraw);
                else
                    hipLaunchKernelGGL((first_layer_raw_kernel<0, 0>), rgrid, dim3(256), rlds, c->stream, (const float*)c->mspec.p, rmin, rtot,
                                       R1[ISS_C_WO], R1[ISS_C_COUT], R1[ISS_C_KH], R1[ISS_C_KW],
                                       (const float*)(n.d_blob + R1[ISS_C_WOFF]), n.kpad[pend], Rraw);
                }
                a.in = Rraw; a.win_row = d_winrow; a.f_rmin = rmin;
                if (pool1) {
                    if (ptot >= (1ll << 40)) return iss_fail(c, ISS_EINVAL, "internal: pooled first layer over %lld items", ptot);
                    float* P = Rraw + (size_t)(rtot + etot) * 4;
                    int32_t* wr2 = reinterpret_cast<int32_t*>(P + (size_t)ptot * 4);
                    hipLaunchKernelGGL(pool_rows_kernel, dim3((unsigned)std::min<long long>((ptot + 255) / 256, 8192)), dim3(256), 0, c->stream,
                                       (const float*)Rraw, P, ptot, (int)rrows, R1[ISS_C_WO], R1[ISS_C_COUT], ph1, pw1, plane_rows);
                    hipLaunchKernelGGL(winrow_pool_kernel, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, c->stream, d_winrow, wr2, bc, rmin, ph1, plane_rows);
                    a.in = P; a.win_row = wr2; a.f_rmin = 0;
                }
                iss_prof_end(c);
                a.stats = d_stats; a.finite = d_fin;
                a.f_bias = n.d_blob + R1[ISS_C_BOFF];
                a.f_wsum = fs1 ? n.d_wsum + n.wsumx_off[pend] : n.d_wsum + n.wsum_off[pend];
                a.f_ps = R1[ISS_C_PSOFF] >= 0 ? n.d_blob + R1[ISS_C_PSOFF] : nullptr;
                a.f_pt = R1[ISS_C_PTOFF] >= 0 ? n.d_blob + R1[ISS_C_PTOFF] : nullptr;
                a.f_act = R1[ISS_C_ACT];
                fl += 2.0 * R1[ISS_C_KH] * R1[ISS_C_KW] * (double)R1[ISS_C_COUT] * (double)bc * R1[ISS_C_HO] * R1[ISS_C_WO];
            }
        }
        ws = ws && fused;
        if (gfused) a.mode = fs1 ? 4 : 3;
        // CHL output through the shared pooled epilogue (conv_common.h epilogue_impl / chl_store): every kernel family that ends in it --
        // the weight-stationary forms, conv_x3_fp_kernel, the generic gather kernel -- can hand its pooled relu output to a
        // conv_x3_wq3h_kernel the way conv_x3_wq_kernel does (the two launch sites with an epilogue of their own take it back below)
        if (x3 && !patch && issk::epi_is_pool_relu(a) && a.Cout % 16 == 0 && want_hl_out(r, a.M / a.pp)) {
            a.out_hl = 1; a.out_np = issk::chl_npad(a.M / a.pp); a.out_f16 = f16mode ? 1 : 0;
            hl_out_row = r; hl_out_np = a.out_np; hl_out_f16 = f16mode;
        }
        auto no_chl_out = [&]() { a.out_hl = 0; a.out_np = 0; a.out_f16 = 0; if (hl_out_row == r) hl_out_row = -1; hl_out_f16 = false; hl_out_dense = false; };
        iss_prof_begin(c, 0, fl);
        iss_prof_tag(c, ws || ws_plain || ws_plain_u || ws_nh2 ? ISS_PROF_WS : fp ? ISS_PROF_FP : !x3 ? ISS_PROF_F32 : ISS_PROF_GATHER);
        iss_prof_row(c, r);
        // one-channel 3x3 'same' first layer of a non-PATCH network: direct f32 kernel (either arithmetic mode)
        const bool no_direct = (c->diag & ISS_DIAG_NO_DIRECT1) != 0;
        const bool direct1 = !no_direct && !patch && pend < 0 && a.Cin == 1 && a.H_k == 3 && a.kw == 3 && a.sh == 1 && a.sw == 1 && a.pt_ == 1 &&
                             a.pl_ == 1 && R[ISS_C_HO] == a.H && R[ISS_C_WO] == a.W && a.pp == 1 && !a.res && !a.ps && a.act <= 1 && a.bias &&
                             a.Cout % 4 == 0 && a.Cout <= 256 && a.M * (long long)(a.Cout / 4) < (1ll << 34);
        if (direct1) {
            iss_prof_tag(c, ISS_PROF_GATHER);
            iss_prof_inst(c, "conv1_direct3x3_kernel<%s>", window ? "true" : "false");
            const long long items = (long long)(a.M / ((long long)a.H * a.W)) * ((a.H + 7) / 8) * a.W * (a.Cout / 4);
            if (items >= (1ll << 32)) return iss_fail(c, ISS_EINVAL, "internal: direct first layer over %lld work items", items);
            const dim3 dgrid((unsigned)std::min<long long>((items + 255) / 256, 8192));
            const size_t dlds = (size_t)9 * a.Cout * sizeof(float);
            if (window) hipLaunchKernelGGL(conv1_direct3x3_kernel<true>, dgrid, dim3(256), dlds, c->stream, a);
            else hipLaunchKernelGGL(conv1_direct3x3_kernel<false>, dgrid, dim3(256), dlds, c->stream, a);
        } else
        if (ws_nh2 && !x3) {
            const unsigned ngroups = (unsigned)((a.M + WS_TM - 1) / WS_TM);
            const unsigned ny = (unsigned)(a.Cout / (2 * BN));
            const dim3 g2(std::min<unsigned>(ngroups, std::max(1u, 256u / ny)), ny);
            const bool trn = a.pp == 1;
            iss_prof_inst(c, "conv_x3_ws_kernel<3,3,false,%s,false,2,1,f32>", trn ? "true" : "false");
            issk::iss_ws_launch_f32_nh2_3x3(a, g2, c->stream, trn);
        } else
        if (ws_nh2) {
            const unsigned ngroups = (unsigned)((a.M + WS_TM - 1) / WS_TM);         // one 512-row tile per group
            const unsigned ny = (unsigned)(a.Cout / (2 * BN));
            const dim3 g2(std::min<unsigned>(ngroups, std::max(1u, 256u / ny)), ny);
            {   // template arguments as iss_ws_launch_nh2_3x3* pick them: <KH,KW,PADDED,TR,FUSED,NH,EPI>
                const bool trn = (padded && !nh2_pad_pool) || (a.pp == 1 && a.Cout % 4 == 0);
                const int epi = padded ? 1 : (trn ? issk::epi_is_simple_tr(a) : issk::epi_is_pool_relu(a));
                iss_prof_inst(c, "conv_x3_ws_kernel<%d,%d,%s,%s,false,2,%d>", a.H_k, a.kw, padded ? "true" : "false", trn ? "true" : "false", epi);
            }
            // one-wave-per-SIMD variant (conv_wq3.h): unpadded, bias + relu (kind 0) or relu + 2 x 1 max-pool (kind 1)
            int wq3_kind = -1;
            if (!(c->diag & ISS_DIAG_NO_WQ) && !padded && a.bias && a.act == 1 && !a.ps && !a.res && a.Cin >= 2 * F2_CH &&
                ws_recip_exact(a.W, issk::WQ3_PIX + a.W)) {
                if (a.pp == 1 && a.M * (long long)a.Cout * 4 < 0xFFF00000ll) wq3_kind = 0;
                else if (a.pp == 2 && a.ph == 2 && a.poolkind == 0 && (a.M / 2) * (long long)a.Cout * 4 < 0xFFF00000ll) wq3_kind = 1;
            }
            if (wq3_kind >= 0) {
                const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 58);
                auto it = n.fp_pix.find(key);
                if (it == n.fp_pix.end()) {
                    int tmr = 0;
                    for (int cand = issk::WQ3_TM; cand >= issk::WQ3_TM - 64 && !tmr; cand -= 4)
                        if (footprint_pixels(a, cand) <= issk::WQ3_PIX) tmr = cand;
                    it = n.fp_pix.emplace(key, tmr).first;
                }
                a.tmr = it->second;
                if (a.tmr <= 0) wq3_kind = -1;
            }
            if (wq3_kind >= 0) {
                const unsigned qtiles = (unsigned)((a.M + a.tmr - 1) / a.tmr);
                const dim3 qgrid(std::min<unsigned>((qtiles + 1) / 2, std::max(1u, 256u / ny)), ny);
                no_chl_out();                                    // (these kernels have epilogues of their own: kind 0 decides below, kind 1 writes f32)
                if (in_is_hl) {                                  // the producer wrote the CHL layout for this launch (want_hl_out)
                    a.in_hl = 1; a.in_np = hl_np[R[ISS_C_IN]];
                    if (hl_f16.count(R[ISS_C_IN])) { a.f16 = 1; a.wh = n.d_wh16 + R[ISS_C_WOFF]; a.wl = n.d_wl16 + R[ISS_C_WOFF]; }
                    if (wq3_kind == 0 && want_hl_out(r, a.M)) { a.out_hl = 1; a.out_np = issk::chl_npad(a.M); hl_out_row = r; hl_out_np = a.out_np; }
                    if (wq3_kind == 1 && want_dhl_out(r, a.Hq, a.Wq)) { a.out_hl = 1; a.out_np = issk::dhl_npad(bc); hl_out_row = r; hl_out_np = a.out_np; hl_out_dense = true; }
                    if (a.out_hl && a.f16) hl_out_f16 = true;
                    iss_prof_inst(c, "conv_x3_wq3h_kernel<%d,%s,%s>", wq3_kind, a.out_hl ? "true" : "false", a.f16 ? "true" : "false");     // <KIND,OUT_HL,F16>
                    issk::iss_wq3h_launch(a, qgrid, c->stream, wq3_kind);
                    in_hl_taken = true;
                } else {
                iss_prof_inst(c, "conv_x3_wq3_kernel<%d>", wq3_kind);
                issk::iss_wq3_launch(a, qgrid, c->stream, wq3_kind);
                }
            } else
            if (padded && nh2_pad_pool) issk::iss_ws_launch_nh2_3x3_padded_pool(a, g2, c->stream);
            else if (padded) issk::iss_ws_launch_nh2_3x3_padded(a, g2, c->stream);
            else issk::iss_ws_launch_nh2_3x3(a, g2, c->stream, a.pp == 1 && a.Cout % 4 == 0);
        } else if (ws_plain_u) {
            const unsigned ngroups = (unsigned)((a.M + (long long)WS_TM * WS_G - 1) / ((long long)WS_TM * WS_G));
            const unsigned per_n = std::max(1u, 256u / grid.y);
            const bool tru = a.pp == 1;
            iss_prof_inst(c, "conv_x3_ws_kernel<3,3,false,%s,false,1,1,plain>", tru ? "true" : "false");
            issk::iss_ws_launch_plain_3x3_unpadded(a, dim3(std::min<unsigned>(ngroups, per_n), grid.y), c->stream, tru);
        } else if (ws_plain) {
            const unsigned ngroups = (unsigned)((a.M + (long long)WS_TM * WS_G - 1) / ((long long)WS_TM * WS_G));
            const unsigned per_n = std::max(1u, 256u / grid.y);                   // one 512-thread workgroup per CU in total
            iss_prof_inst(c, "conv_x3_ws_kernel<3,3,true,true,false,1,%d>", (int)issk::epi_is_simple_tr(a));
            issk::iss_ws_launch_plain_3x3(a, dim3(std::min<unsigned>(ngroups, per_n), grid.y), c->stream);
        } else if (ws && ws_f32) {
            const unsigned ngroups = (unsigned)((a.M + (long long)WS_TM * WS_G - 1) / ((long long)WS_TM * WS_G));
            const dim3 wgrid(std::min<unsigned>(ngroups, 256u), grid.y);
            iss_prof_inst(c, "conv_x3_ws_kernel<5,3,false,false,true,1,1,f32>");
            issk::iss_ws_launch_f32_fused_5x3(a, wgrid, c->stream);
        } else if (ws && ws_ring) {
            const unsigned ngroups = (unsigned)((a.M + a.tmr - 1) / a.tmr);      // one tile of tmr rows per group
            const dim3 wgrid(std::min<unsigned>(ngroups, 256u), grid.y);
            iss_prof_inst(c, "conv_x3_ws_kernel<%d,%d,%s,false,true,1,%d,ring>", a.H_k, a.kw, padded ? "true" : "false", (int)issk::epi_is_pool_relu_any(a));
            issk::iss_ws_launch_ring(a, wgrid, c->stream, padded);
        } else if (ws && ws_fs) {
            const unsigned ngroups = (unsigned)((a.M + (long long)WS_TM - 1) / (long long)WS_TM);      // FS: one tile per group (conv_ws.h G)
            const dim3 wgrid(std::min<unsigned>(ngroups, 256u), grid.y);
            iss_prof_inst(c, "conv_x3_ws_kernel<%d,%d,%s,%s,true,1,%d,fs>", a.H_k, a.kw, padded ? "true" : "false", fs_tr_ok ? "true" : "false",
                          fs_tr_ok ? 1 : (int)issk::epi_is_pool_relu_any(a));
            if (fs_tr_ok) issk::iss_ws_launch_fs_5x3_tr(a, wgrid, c->stream);
            else if (a.H_k == 5) issk::iss_ws_launch_fs_5x3(a, wgrid, c->stream, padded);
            else issk::iss_ws_launch_fs_3x3(a, wgrid, c->stream, padded);
        } else if (ws) {
            const unsigned ngroups = (unsigned)((a.M + (long long)WS_TM * WS_G - 1) / ((long long)WS_TM * WS_G));
            const dim3 wgrid(std::min<unsigned>(ngroups, 256u), grid.y);         // persistent: one 512-thread workgroup per CU
            const bool tr = a.pp == 1 && a.Cout % 4 == 0;
            // <= 32 output channels: one 32-column block per workgroup (conv_ws.h NCB = 1) instead of half-empty 64-column ones
            const bool ncb1 = !(c->diag & ISS_DIAG_NO_NCB1) && fused && !fs1 && !padded && !tr && a.Cout <= 32 && issk::epi_is_pool_relu(a) &&
                              issk::iss_ws_ncb1_compiled(a.H_k, a.kw) && a.sh == 1 && a.sw == 1 && a.Cin >= 2 * F2_CH;
            if (ncb1) {
                iss_prof_inst(c, "conv_x3_ws_kernel<%d,%d,false,false,true,1,1,ncb1>", a.H_k, a.kw);
                issk::iss_ws_launch_ncb1_5x3(a, wgrid, c->stream);
            } else {
            // one-wave-per-SIMD, two-footprint variant (conv_wq.h): the dominant launch of the segmenter nets
            bool wq = false;
            if (!(c->diag & ISS_DIAG_NO_WQ) && fused && !padded && !tr && issk::epi_is_pool_relu(a) && a.pp == 4 && a.ph == 2 &&
                issk::iss_wq_compiled(a.H_k, a.kw) && a.sh == 1 && a.sw == 1 && a.Cin >= 2 * F2_CH && a.M % 4 == 0 &&
                (a.M / 4) * (long long)a.Cout * 4 < 0xFFF00000ll && a.Cout <= 256) {
                // rows per tile: the largest multiple of 4 (<= 512) whose footprint fits the kernel's 800 pixels
                const long long key = ((long long)r << 32) | (unsigned)bc | (1ll << 59);
                auto it = n.fp_pix.find(key);
                if (it == n.fp_pix.end()) {
                    int tmr = 0;
                    for (int cand = WS_TM; cand >= WS_TM - 32 && !tmr; cand -= 4)
                        if (footprint_pixels(a, cand) <= issk::WQ_PIX) tmr = cand;
                    it = n.fp_pix.emplace(key, tmr).first;
                }
                a.tmr = it->second;
                wq = a.tmr > 0;
            }
            if (wq) {
                const unsigned qtiles = (unsigned)((a.M + a.tmr - 1) / a.tmr);
                const dim3 qgrid(std::min<unsigned>((qtiles + 1) / 2, 256u), grid.y);     // persistent: one 256-thread workgroup per CU
                no_chl_out();                                    // (its own CHL epilogue; the halves follow the launch's operand type)
                if (a.Cout % BN == 0 && want_hl_out(r, a.M / 4)) { a.out_hl = 1; a.out_np = issk::chl_npad(a.M / 4); hl_out_row = r; hl_out_np = a.out_np; }
                if (row_f16) { a.f16 = 1; a.wh = n.d_wh16 + R[ISS_C_WOFF]; a.wl = n.d_wl16 + R[ISS_C_WOFF]; if (a.out_hl) hl_out_f16 = true; }
                iss_prof_inst(c, "conv_x3_wq_kernel<%d,%d,%s,%s>", a.H_k, a.kw, a.out_hl ? "true" : "false", a.f16 ? "true" : "false");   // <KH,KW,OUT_HL,F16>
                issk::iss_wq_launch_5x3(a, qgrid, c->stream);
            } else {
            {
                const int epi = tr ? issk::epi_is_simple_tr(a) : issk::epi_is_pool_relu_any(a);
                iss_prof_inst(c, "conv_x3_ws_kernel<%d,%d,%s,%s,true,1,%d>", a.H_k, a.kw, padded ? "true" : "false", tr ? "true" : "false", epi);
            }
#define ISS_WS_CASE(KH_, KW_) if (a.H_k == KH_ && a.kw == KW_) iss_ws_launch_##KH_##x##KW_(a, wgrid, c->stream, padded, tr, fused); else
            ISS_WS_SHAPES(ISS_WS_CASE) { return iss_fail(c, ISS_EINVAL, "internal: no weight-stationary kernel for %dx%d", a.H_k, a.kw); }
#undef ISS_WS_CASE
            }
            }
        } else if (fp) {
#define ISS_FP_CASE(KH_, KW_) if (a.H_k == KH_ && a.kw == KW_) iss_fp_launch_##KH_##x##KW_(a, pgrid, c->stream, padded, tr, fused, nh); else
            // 128 output channels per workgroup where the layer has them: one LDS footprint serves two 64-column halves
            const bool no_nh2 = (c->diag & ISS_DIAG_NO_NH2) != 0;
            const int nh = (!fused && !no_nh2 && issk::iss_fp_has_nh2(a.H_k, a.kw) && a.Cout % (2 * BN) == 0) ? 2 : 1;
            const dim3 pgrid(std::min<unsigned>(a.nblk, 512u), grid.y / nh);     // persistent: 2 workgroups per CU
            const bool no_tr = (c->diag & ISS_DIAG_NO_TR) != 0;             // diagnostic: row-major epilogue everywhere
            const bool tr = !no_tr && a.pp == 1 && a.Cout % 4 == 0;         // float4 epilogue on transposed accumulators
            if (fused && a.H_k * a.kw >= 12) iss_prof_inst(c, "conv_x3_fp_kernel<%d,%d,false,%s,true,1>", a.H_k, a.kw, tr ? "true" : "false");
            else iss_prof_inst(c, "conv_x3_fp_kernel<%d,%d,%s,%s,false,%d>", a.H_k, a.kw, padded ? "true" : "false", tr ? "true" : "false", nh);
            ISS_FP_SHAPES(ISS_FP_CASE) { return iss_fail(c, ISS_EINVAL, "internal: no footprint kernel for %dx%d", a.H_k, a.kw); }
#undef ISS_FP_CASE
        } else if (x3 && patch && a.H_k * a.kw <= XBK && a.M < (1ll << 31)) {
            const dim3 pgrid(std::min<unsigned>(a.nblk, 512u), grid.y);     // persistent, no barriers: 2 workgroups per CU
            iss_prof_tag(c, ISS_PROF_PATCH1);
            iss_prof_inst(c, "conv1_patch_x3_kernel<%s>", (a.pp == 1 && a.Cout % 4 == 0 && !a.res) ? "true" : "false");
            // blob offsets are multiples of 8 floats, so the float4 loads of bias / scale / shift are aligned
            if (a.pp == 1 && a.Cout % 4 == 0 && !a.res) hipLaunchKernelGGL(conv1_patch_x3_kernel<true>, pgrid, dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL(conv1_patch_x3_kernel<false>, pgrid, dim3(256), 0, c->stream, a);
        } else if (x3) {
            const bool tr = a.pp == 1 && a.Cout % 4 == 0;     // float4 epilogue on transposed accumulators
            // wider N tiles for wide layers (A staged once per 128 / 256 output channels); 1-D XCD-aware grid
            // NTN = 4 (128 x 128 tiles) is compiled but not selected: with the XCD-aware order the A tile is re-read from
            // L2, not from HBM, and the wider tiles (fewer, fatter workgroups) measured 6 % SLOWER on ResNet-101
            const int ntn = 2;
            a.nblk_n = (unsigned)((a.Cout + 32 * ntn - 1) / (32 * ntn));
            const dim3 gridw(a.nblk * a.nblk_n);
            const bool no_pw = (c->diag & ISS_DIAG_NO_PW) != 0;
            if (gfused) {
                iss_prof_inst(c, "conv_x3_kernel<%d,%s,2>", a.mode, tr ? "true" : "false");
                if (a.mode == 4 && tr) hipLaunchKernelGGL((conv_x3_kernel<4, true, 2>), gridw, dim3(256), 0, c->stream, a);
                else if (a.mode == 4) hipLaunchKernelGGL((conv_x3_kernel<4, false, 2>), gridw, dim3(256), 0, c->stream, a);
                else if (tr) hipLaunchKernelGGL((conv_x3_kernel<3, true, 2>), gridw, dim3(256), 0, c->stream, a);
                else hipLaunchKernelGGL((conv_x3_kernel<3, false, 2>), gridw, dim3(256), 0, c->stream, a);
                iss_prof_end(c);
                return ISS_OK;
            }
            const bool pointwise = !no_pw && a.mode == 0 && tr && a.H_k == 1 && a.kw == 1 && a.sh == 1 && a.sw == 1 && a.pt_ == 0 &&
                                   a.pl_ == 0 && R[ISS_C_HO] == a.H && R[ISS_C_WO] == a.W && a.Kpad == a.Cin;
            if (pointwise) iss_prof_tag(c, ISS_PROF_PW);
            // Streaming pointwise kernels (conv_pw.h) for K <= 2048.  The segmenter nets' first dense layer (K = 4992 / 8320,
            // 192 columns, ~28 k rows per launch) keeps conv_x3_pw_kernel: it runs at 1.7 TB/s of activations on every tiling
            // that was built for it (deeper ring -8 %; one workgroup per 64 rows x all 192 columns +6 %, with split-K +3..+11 %,
            // with non-temporal activation loads +8 %: profiles/HISTORY.md, round 3)
#ifdef ISS_PW_NO_ASM_RING                        // build-time escape when tools/check_ring_regs.py rejects this compiler's cnn_pw.o (Makefile)
            const bool no_pws = true;
#else
            const bool no_pws = (c->diag & ISS_DIAG_NO_PWS) != 0;                // diagnostic: the round-2 pointwise kernel everywhere
#endif
            const bool no_pws2 = (c->diag & ISS_DIAG_NO_PWS2) != 0;              // diagnostic: 64-column tiles everywhere
            const bool pws_ok = !no_pws && a.Kpad <= 2048 && !f16_dense_pw;        // (fp16 mode: a dense layer that would otherwise run in exact f32)
            // strided 1x1 (the shortcut projections): the 128-column kernel on a strided pixel list
            const bool pw_strided = pws_ok && !no_pws2 && a.mode == 0 && tr && a.H_k == 1 && a.kw == 1 && (a.sh > 1 || a.sw > 1) && a.pt_ == 0 &&
                                    a.pl_ == 0 && a.Kpad == a.Cin && issk::pws2_strided_supported(a, R[ISS_C_HO], R[ISS_C_WO]);
            const bool simple_pw = a.act <= 1 && !a.ps;
            if (pw_strided) { iss_prof_tag(c, ISS_PROF_PW); iss_prof_inst(c, "conv_x3_pws2_kernel<true,true>"); issk::iss_pws2_launch(a, c->stream, true); }
            else if (pointwise && pws_ok && !no_pws2 && issk::pws2_supported(a)) {
                iss_prof_inst(c, "conv_x3_pws2_kernel<%s,false>", simple_pw ? "true" : "false");
                issk::iss_pws2_launch(a, c->stream);
            } else if (pointwise && pws_ok && issk::pws_supported(a)) {
                iss_prof_inst(c, "conv_x3_pws_kernel<%s,%s>", a.res ? "true" : "false", simple_pw ? "true" : "false");
                issk::iss_pws_launch(a, dim3(std::min<unsigned>(a.nblk * a.nblk_n, 512u)), c->stream);
            } else if (pointwise) {
                if (row_f16) {
                    a.f16 = 1; a.wh = n.d_wh16 + R[ISS_C_WOFF]; a.wl = n.d_wl16 + R[ISS_C_WOFF];
                    iss_prof_inst(c, "conv_x3_pw_kernel<true>");                                  // <F16>
                    hipLaunchKernelGGL(conv_x3_pw_kernel<true>, dim3(std::min<unsigned>(a.nblk * a.nblk_n, 768u)), dim3(256), 0, c->stream, a);
                } else {
                iss_prof_inst(c, "conv_x3_pw_kernel<false>");
                hipLaunchKernelGGL(conv_x3_pw_kernel<false>, dim3(std::min<unsigned>(a.nblk * a.nblk_n, 768u)), dim3(256), 0, c->stream, a);
                }
            } else {
            iss_prof_inst(c, "conv_x3_kernel<%d,%s,2>", a.mode, (a.mode != 2 && tr) ? "true" : "false");
            if (ntn == 4) hipLaunchKernelGGL((conv_x3_kernel<0, true, 4>), gridw, dim3(256), 0, c->stream, a);
            else if (a.mode == 0 && tr) hipLaunchKernelGGL((conv_x3_kernel<0, true, 2>), gridw, dim3(256), 0, c->stream, a);
            else if (a.mode == 0) hipLaunchKernelGGL((conv_x3_kernel<0, false, 2>), gridw, dim3(256), 0, c->stream, a);
            else if (a.mode == 1 && tr) hipLaunchKernelGGL((conv_x3_kernel<1, true, 2>), gridw, dim3(256), 0, c->stream, a);
            else if (a.mode == 1) hipLaunchKernelGGL((conv_x3_kernel<1, false, 2>), gridw, dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL((conv_x3_kernel<2, false, 2>), gridw, dim3(256), 0, c->stream, a);
            }
        } else {
            iss_prof_inst(c, "conv_igemm_kernel<%d>", a.mode);
            if (a.mode == 0) hipLaunchKernelGGL(conv_igemm_kernel<0>, grid1, dim3(256), 0, c->stream, a);
            else if (a.mode == 1) hipLaunchKernelGGL(conv_igemm_kernel<1>, grid1, dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL(conv_igemm_kernel<2>, grid1, dim3(256), 0, c->stream, a);
        }
        iss_prof_end(c);
        if (in_is_hl && !in_hl_taken) return iss_fail(c, ISS_EINVAL, "internal: row %d reads a CHL tensor on a kernel that expects f32", r);
        return ISS_OK;
    };
    int pending = -1;                                            // deferred PATCH first layer (see can_defer)
    for (int r = 0; r < n.nrows; ++r) {
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        const float* in = R[ISS_C_IN] == ISS_BUF_INPUT ? d_input : (const float*)c->act[R[ISS_C_IN]].p;
        float* out = (float*)c->act[R[ISS_C_OUT]].p;
        const int op = R[ISS_C_OP];
        if (op == ISS_OP_CONV) {
            if (pending < 0 && can_defer(r)) { pending = r; *result = out; continue; }
            // identity-residual expansion followed by the next block's reduction: one chained launch (conv_pwc.h)
            if (asm_ring_ok && pending < 0 && x3mode && chain_pair(r) &&
                !(c->diag & (ISS_DIAG_NO_CHAIN | ISS_DIAG_NO_PW | ISS_DIAG_NO_PWS | ISS_DIAG_NO_PWS2))) {
                const int rc2 = conv_row(r, -1, -1, r + 1);
                if (rc2 == ISS_OK) {
                    ISS_HIP(c, hipGetLastError());
                    hl_np.erase(R[ISS_C_OUT]);
                    ++r;
                    hl_np.erase(n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OUT]);
                    *result = (float*)c->act[n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OUT]].p;
                    continue;
                }
                if (rc2 != kDualDeclined) return rc2;
            }
            // projection shortcut followed by its expansion (ISS_C_DUALW on the next row): one two-source launch when the split-bf16
            // streaming kernels are in use (the diagnostic switches that move 1x1 layers elsewhere keep their meaning)
            if (asm_ring_ok && pending < 0 && x3mode && r + 1 < n.nrows && n.prog[(size_t)(r + 1) * ISS_PROG_COLS + ISS_C_DUALW] > 0 &&
                !(c->diag & (ISS_DIAG_NO_DUAL | ISS_DIAG_NO_PW | ISS_DIAG_NO_PWS | ISS_DIAG_NO_PWS2))) {
                const int rc2 = conv_row(r + 1, -1, r, -1);
                if (rc2 == ISS_OK) {
                    ISS_HIP(c, hipGetLastError());
                    hl_np.erase(R[ISS_C_OUT]);
                    ++r;
                    hl_np.erase(n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OUT]);
                    *result = (float*)c->act[n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OUT]].p;
                    continue;
                }
                if (rc2 != kDualDeclined) return rc2;
            }
            const int rc = conv_row(r, pending, -1, -1);
            if (pending >= 0) hl_np.erase(n.prog[(size_t)pending * ISS_PROG_COLS + ISS_C_OUT]);
            pending = -1;
            if (rc) return rc;
        } else if (op == ISS_OP_POOL) {
            const long long total = (long long)bc * R[ISS_C_HO] * R[ISS_C_WO] * R[ISS_C_CIN];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               total, R[ISS_C_H], R[ISS_C_W], R[ISS_C_CIN], R[ISS_C_HO], R[ISS_C_WO], R[ISS_C_KH],
                               R[ISS_C_KW], R[ISS_C_SH], R[ISS_C_SW], R[ISS_C_PT], R[ISS_C_PL], R[ISS_C_POOLKIND]);
            iss_prof_end(c);
        } else if (op == ISS_OP_SOFTMAX) {
            const long long rows = (long long)bc * R[ISS_C_H] * R[ISS_C_W];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               rows, R[ISS_C_CIN]);
            iss_prof_end(c);
        } else if (op == ISS_OP_ACT) {
            const long long total = (long long)bc * R[ISS_C_H] * R[ISS_C_W] * R[ISS_C_CIN];
            float alpha, p2, p3;
            { const int32_t bits = R[ISS_C_ACTPARAM]; memcpy(&alpha, &bits, sizeof(float)); }
            { const int32_t bits = R[ISS_C_ACTPARAM2]; memcpy(&p2, &bits, sizeof(float)); }
            { const int32_t bits = R[ISS_C_ACTPARAM3]; memcpy(&p3, &bits, sizeof(float)); }
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(act_kernel, dim3((unsigned)std::min<long long>((total + 1023) / 1024, 1 << 20)), dim3(256), 0, c->stream, in, out, total,
                               R[ISS_C_ACT], alpha, p2, p3);
            iss_prof_end(c);
        } else if (op == ISS_OP_ELT) {
            const int k = R[ISS_C_ACT];
            const long long hw = (long long)R[ISS_C_H] * R[ISS_C_W];
            const bool chan = k == ISS_ELT_COPY || k == ISS_ELT_ZERO;
            const long long total = (long long)bc * hw * (chan ? R[ISS_C_KH] : R[ISS_C_CIN]);
            const float* res = R[ISS_C_RES] >= 0 ? (const float*)c->act[R[ISS_C_RES]].p : nullptr;
            if (res && hl_np.count(R[ISS_C_RES])) return iss_fail(c, ISS_EINVAL, "internal: row %d reads a CHL tensor", r);
            int st[3] = {0, 0, 0};
            if (k == ISS_ELT_PERMUTE) {
                const int sin[3] = {R[ISS_C_W] * R[ISS_C_CIN], R[ISS_C_CIN], 1};
                st[0] = sin[R[ISS_C_KH]]; st[1] = sin[R[ISS_C_KW]]; st[2] = sin[R[ISS_C_SH]];
            }
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(elt_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 1 << 20)), dim3(256), 0, c->stream, in, res, out,
                               total, k, R[ISS_C_CIN], R[ISS_C_COUT], R[ISS_C_KH], R[ISS_C_PT], R[ISS_C_PL],
                               R[ISS_C_HO], R[ISS_C_WO], R[ISS_C_COUT], st[0], st[1], st[2], R[ISS_C_ORDER] == 1 ? 1 : 0);
            iss_prof_end(c);
        } else if (op == ISS_OP_STATPOOL) {
            const long long total = (long long)bc * R[ISS_C_H] * R[ISS_C_CIN];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(statpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               total, R[ISS_C_H], R[ISS_C_W], R[ISS_C_CIN]);
            iss_prof_end(c);
        }
        ISS_HIP(c, hipGetLastError());
        *result = out;
        if (hl_out_row == r) {
            hl_np[R[ISS_C_OUT]] = hl_out_np;
            if (hl_out_f16) hl_f16[R[ISS_C_OUT]] = true; else hl_f16.erase(R[ISS_C_OUT]);
            if (hl_out_dense) hl_dense[R[ISS_C_OUT]] = true; else hl_dense.erase(R[ISS_C_OUT]);
        } else { hl_np.erase(R[ISS_C_OUT]); hl_f16.erase(R[ISS_C_OUT]); hl_dense.erase(R[ISS_C_OUT]); }
        hl_out_f16 = false; hl_out_dense = false;
        if (op != ISS_OP_CONV && R[ISS_C_IN] != ISS_BUF_INPUT && hl_np.count(R[ISS_C_IN]))
            return iss_fail(c, ISS_EINVAL, "internal: row %d reads a CHL tensor", r);
    }
    return ISS_OK;
}

int plan_chunk(iss_ctx* c, IssNet& n, int total, int* bc_out) {
    int64_t per = 0;
    for (auto e : n.buf_elems) per += e;
    int64_t bc = (int64_t)(c->ws_limit / (uint64_t)(per * sizeof(float)));
    if (bc < 1) bc = 1;
    if (bc > total) bc = total;
    // keep every per-buffer float index below 2^31: several kernels form element offsets in 32 bits (first_layer_raw_kernel's
    // `(unsigned)total`, the footprint kernels' row * Cout offsets), whatever workspace limit the caller has set
    int64_t emax = 1;
    for (auto e : n.buf_elems) emax = std::max<int64_t>(emax, e);
    bc = std::min<int64_t>(bc, ((1ll << 31) - 1) / emax);
    if (bc < 1) return iss_fail(c, ISS_EINVAL, "network activation of %lld floats per sample exceeds the 2^31 element limit", (long long)emax);
    if (bc > 8) bc -= bc % 8;
    if ((int)c->act.size() < n.nbuf) c->act.resize(n.nbuf);
    for (int i = 0; i < n.nbuf; ++i) {
        int rc = iss_reserve(c, c->act[i], (size_t)bc * n.buf_elems[i] * sizeof(float) + issk::ISS_ACT_SLACK);
        if (rc) return rc;
    }
    *bc_out = (int)bc;
    return ISS_OK;
}

}  // namespace

static int cnn_probs_impl(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots, float* probs_out,
                          uint8_t* finite_out, bool async, int64_t* ticket_out);

// Precision guard (include/iss.h): the network's first call in split-bf16 mode.  Up to 256 of the call's windows -- four runs of
// consecutive slots at the quarters of the list, so that each run takes the kernels the call itself will take (the shared first
// layer needs overlapping windows) -- are evaluated in both arithmetic modes; max |d log p| decides.
static int precision_guard(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots) {
    IssNet& n = c->nets[id];
    const int eff = n.prec_override >= 0 ? n.prec_override : c->precision;
    if (eff == ISS_PREC_F32) { n.guard_state = ISS_GUARD_FIXED; return ISS_OK; }
    if (!(c->guard_threshold > 0.f)) return ISS_OK;               // guard off: stays pending
    const int runs = nslots >= 256 ? 4 : 1, per = nslots >= 256 ? 64 : nslots;
    const size_t od = (size_t)n.out_dim;
    std::vector<float> pref((size_t)runs * per * od), pm((size_t)per * od);
    std::vector<uint8_t> fref((size_t)runs * per), fm(per);
    auto window_run = [&](int r) { return win_row + (runs == 1 ? 0 : (size_t)r * (nslots - per) / (runs - 1)); };
    int rc = ISS_OK, compared = 0;
    // max |log p_mode - log p_f32| over every class of every finite window (non-finite ones carry the constant 0.5, segmenter.py:175);
    // a NaN (an activation beyond fp16's range) counts as a failure; classes below 1e-30 in either mode are skipped
    auto probe = [&](int mode, double& worst) {
        worst = 0.0; compared = 0;
        for (int r = 0; r < runs && rc == ISS_OK; ++r) {
            n.prec_override = mode;
            rc = cnn_probs_impl(c, id, window_run(r), per, pm.data(), fm.data(), false, nullptr);
            if (rc != ISS_OK) return;
            for (int i = 0; i < per; ++i) {
                if (!fm[i] || !fref[(size_t)r * per + i]) continue;
                ++compared;
                for (size_t k = 0; k < od; ++k) {
                    const float a = pm[(size_t)i * od + k], b = pref[((size_t)r * per + i) * od + k];
                    if (!(a == a)) { worst = 1e30; continue; }
                    if (!(a > 1e-30f) || !(b > 1e-30f)) continue;      // (an underflowing class: a real disagreement shows in the window's other classes)
                    worst = std::max(worst, std::fabs(std::log((double)a) - std::log((double)b)));
                }
            }
        }
    };
    c->in_guard = true;
    for (int r = 0; r < runs && rc == ISS_OK; ++r) {              // the reference: exact f32
        n.prec_override = ISS_PREC_F32;
        rc = cnn_probs_impl(c, id, window_run(r), per, pref.data() + (size_t)r * per * od, fref.data() + (size_t)r * per, false, nullptr);
    }
    double worst = 0.0;
    int chosen = eff;
    if (rc == ISS_OK) probe(eff, worst);
    const double first = worst;
    if (rc == ISS_OK && worst > (double)c->guard_threshold) {
        chosen = ISS_PREC_F32;
        // the other split mode first (the same speed as the one that failed): fp16 halves where bf16 ones are too coarse, bf16 halves
        // where an activation left fp16's range
        const int other = eff == ISS_PREC_BF16X3 ? ISS_PREC_F16X3 : ISS_PREC_BF16X3;
        if (other == ISS_PREC_BF16X3 || n.f16_ok) {
            double w2 = 0.0;
            probe(other, w2);
            if (rc == ISS_OK && w2 <= (double)c->guard_threshold) { chosen = other; worst = w2; }
        }
    }
    c->in_guard = false;
    n.prec_override = -1;
    if (rc != ISS_OK) return rc;
    n.guard_dlogp = (float)std::min(first, 1e30); n.guard_slots = compared;
    n.guard_dlogp_chosen = chosen == eff ? n.guard_dlogp : (chosen == ISS_PREC_F32 ? 0.f : (float)worst);
    if (chosen != eff) { n.prec_override = chosen; n.guard_state = ISS_GUARD_ESCALATED; }
    else n.guard_state = ISS_GUARD_PASSED;
    return ISS_OK;
}

static int cnn_probs_impl(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots, float* probs_out,
                          uint8_t* finite_out, bool async, int64_t* ticket_out) {
    if (!c) return ISS_EINVAL;
    if (ticket_out) *ticket_out = -1;
    if (id < 0 || id >= ISS_MAX_NETS || nslots < 0 || (nslots > 0 && (!win_row || !probs_out || !finite_out)))
        return iss_fail(c, ISS_EINVAL, "iss_cnn_probs: bad argument");
    IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "iss_cnn_probs: net %d not loaded", id);
    if (!c->have_feats) return iss_fail(c, ISS_ESTATE, "iss_cnn_probs: no mel spectrogram resident");
    if (n.in_h != 68 || n.in_c != 1 || n.in_w > 24) return iss_fail(c, ISS_EINVAL, "net %d is not a (68,h,1) patch network", id);
    if (nslots == 0) return ISS_OK;
    for (int i = 0; i < nslots; ++i)
        if (win_row[i] < 0 || win_row[i] + 68 > c->T)
            return iss_fail(c, ISS_EINVAL, "iss_cnn_probs: window %d (row %d) outside the %d resident frames", i, win_row[i], c->T);
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if (!c->in_guard && n.guard_state == ISS_GUARD_PENDING && (rc = precision_guard(c, id, win_row, nslots))) return rc;
    if ((rc = iss_reserve(c, c->d_winrow, (size_t)nslots * 4))) return rc;
    if ((rc = iss_reserve(c, c->d_stats, (size_t)nslots * 8))) return rc;
    if ((rc = iss_reserve(c, c->d_finite, (size_t)nslots))) return rc;
    if ((rc = iss_reserve(c, c->d_out, (size_t)nslots * n.out_dim * 4))) return rc;
    int bc = 0;
    if ((rc = plan_chunk(c, n, nslots, &bc))) return rc;
    if (async) {                                     // the caller may reuse win_row as soon as we return: stage it (pinned)
        void* pinned; int slot;
        if ((rc = iss_stage_host(c, win_row, (size_t)nslots * 4, &pinned, &slot))) return rc;
        ISS_HIP(c, hipMemcpyAsync(c->d_winrow.p, pinned, (size_t)nslots * 4, hipMemcpyHostToDevice, c->stream));
        iss_stage_mark(c, slot);
    } else {
        ISS_HIP(c, hipMemcpyAsync(c->d_winrow.p, win_row, (size_t)nslots * 4, hipMemcpyHostToDevice, c->stream));
    }
    iss_prof_begin(c, 2, 0);
    hipLaunchKernelGGL(patch_stats_kernel, dim3((nslots + 3) / 4), dim3(256), 0, c->stream, (const float*)c->mspec.p,
                       (const int32_t*)c->d_winrow.p, nslots, n.in_w, (float*)c->d_stats.p, (uint8_t*)c->d_finite.p);
    iss_prof_end(c);
    ISS_HIP(c, hipGetLastError());
    // Shared first layer (ConvArgs::f_*): decided per call from the whole window list, not per chunk, so that the result
    // does not depend on the workspace limit: on when the windows overlap at least 4-fold on average.
    bool share = !(c->diag & ISS_DIAG_NO_SHARED_FIRST);
    {
        int gmin = win_row[0], gmax = win_row[0];
        for (int i = 1; i < nslots; ++i) { gmin = std::min(gmin, win_row[i]); gmax = std::max(gmax, win_row[i]); }
        if ((long long)(gmax - gmin + 68) * 4 > (long long)nslots * 68) share = false;
    }
    for (int s0 = 0; s0 < nslots; s0 += bc) {
        const int cur = std::min(bc, nslots - s0);
        float* res = nullptr;
        int rmin = win_row[s0], rmax = win_row[s0];
        for (int i = s0 + 1; i < s0 + cur; ++i) { rmin = std::min(rmin, win_row[i]); rmax = std::max(rmax, win_row[i]); }
        rc = run_program(c, n, cur, (const int32_t*)c->d_winrow.p + s0, (const float*)c->d_stats.p + 2 * (size_t)s0,
                         (const uint8_t*)c->d_finite.p + s0, nullptr, &res, rmin, rmax, share);
        if (rc) return rc;
        ISS_HIP(c, hipMemcpyAsync((float*)c->d_out.p + (size_t)s0 * n.out_dim, res, (size_t)cur * n.out_dim * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    const long long tot = (long long)nslots * n.out_dim;
    hipLaunchKernelGGL(mask_probs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                       (float*)c->d_out.p, (const uint8_t*)c->d_finite.p, (long long)nslots, n.out_dim);
    ISS_HIP(c, hipGetLastError());
    ISS_HIP(c, hipMemcpyAsync(probs_out, c->d_out.p, (size_t)tot * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipMemcpyAsync(finite_out, c->d_finite.p, (size_t)nslots, hipMemcpyDeviceToHost, c->stream));
    if (async) {
        hipEvent_t ev;
        if (!c->ev_pool.empty()) { ev = c->ev_pool.back(); c->ev_pool.pop_back(); }
        else ISS_HIP(c, hipEventCreate(&ev));
        ISS_HIP(c, hipEventRecord(ev, c->stream));
        c->ticket_ev.push_back(ev);
        if (ticket_out) *ticket_out = c->ticket_base + (int64_t)c->ticket_ev.size() - 1;
        return ISS_OK;
    }
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}

extern "C" int iss_cnn_probs(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots, float* probs_out,
                             uint8_t* finite_out) {
    return cnn_probs_impl(c, id, win_row, nslots, probs_out, finite_out, false, nullptr);
}

extern "C" int iss_cnn_probs_async(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots, float* probs_out,
                                   uint8_t* finite_out, int64_t* ticket_out) {
    return cnn_probs_impl(c, id, win_row, nslots, probs_out, finite_out, true, ticket_out);
}

extern "C" int iss_cnn_forward(iss_ctx* c, int id, const float* x, int32_t nsamp, float* out) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || nsamp < 0 || (nsamp > 0 && (!x || !out)))
        return iss_fail(c, ISS_EINVAL, "iss_cnn_forward: bad argument");
    IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "iss_cnn_forward: net %d not loaded", id);
    for (int r = 0; r < n.nrows; ++r)
        if (n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OP] == ISS_OP_CONV && n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_INMODE] == 1)
            return iss_fail(c, ISS_EINVAL, "iss_cnn_forward: net %d reads mspec patches; use iss_cnn_probs", id);
    if (nsamp == 0) return ISS_OK;
    ISS_HIP(c, hipSetDevice(c->device));
    const size_t in_elems = (size_t)n.in_h * n.in_w * n.in_c;
    int rc, bc = 0;
    if ((rc = plan_chunk(c, n, nsamp, &bc))) return rc;
    if ((rc = iss_reserve(c, c->d_in, (size_t)bc * in_elems * 4))) return rc;
    if ((rc = iss_reserve(c, c->d_out, (size_t)nsamp * n.out_dim * 4))) return rc;
    for (int s0 = 0; s0 < nsamp; s0 += bc) {
        const int cur = std::min(bc, nsamp - s0);
        ISS_HIP(c, hipMemcpyAsync(c->d_in.p, x + (size_t)s0 * in_elems, (size_t)cur * in_elems * 4, hipMemcpyHostToDevice, c->stream));
        float* res = nullptr;
        rc = run_program(c, n, cur, nullptr, nullptr, nullptr, (const float*)c->d_in.p, &res);
        if (rc) return rc;
        ISS_HIP(c, hipMemcpyAsync((float*)c->d_out.p + (size_t)s0 * n.out_dim, res, (size_t)cur * n.out_dim * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    ISS_HIP(c, hipMemcpyAsync(out, c->d_out.p, (size_t)nsamp * n.out_dim * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}

extern "C" int iss_vbx_embed(iss_ctx* c, int id, const int32_t* starts, int32_t nwin, float* out) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || nwin < 0 || (nwin > 0 && (!starts || !out)))
        return iss_fail(c, ISS_EINVAL, "iss_vbx_embed: bad argument");
    IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "iss_vbx_embed: net %d not loaded", id);
    if (n.prog[ISS_C_OP] != ISS_OP_CONV || n.prog[ISS_C_INMODE] != 2)
        return iss_fail(c, ISS_EINVAL, "iss_vbx_embed: net %d does not start with a window-mode conv", id);
    if (c->vbx_T <= 0) return iss_fail(c, ISS_ESTATE, "iss_vbx_embed: no vbx features resident (iss_vbx_features*)");
    if (nwin == 0) return ISS_OK;
    for (int i = 0; i < nwin; ++i)
        if (starts[i] < 0 || starts[i] + n.in_w > c->vbx_T)
            return iss_fail(c, ISS_EINVAL, "iss_vbx_embed: window %d (frames %d..%d) outside the %d resident frames", i, starts[i],
                            starts[i] + n.in_w, c->vbx_T);
    ISS_HIP(c, hipSetDevice(c->device));
    int rc, bc = 0;
    if ((rc = iss_reserve(c, c->d_winrow, (size_t)nwin * 4))) return rc;
    if ((rc = iss_reserve(c, c->d_out, (size_t)nwin * n.out_dim * 4))) return rc;
    if ((rc = plan_chunk(c, n, nwin, &bc))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->d_winrow.p, starts, (size_t)nwin * 4, hipMemcpyHostToDevice, c->stream));
    for (int s0 = 0; s0 < nwin; s0 += bc) {
        const int cur = std::min(bc, nwin - s0);
        float* res = nullptr;
        rc = run_program(c, n, cur, (const int32_t*)c->d_winrow.p + s0, nullptr, nullptr, nullptr, &res);
        if (rc) return rc;
        ISS_HIP(c, hipMemcpyAsync((float*)c->d_out.p + (size_t)s0 * n.out_dim, res, (size_t)cur * n.out_dim * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    ISS_HIP(c, hipMemcpyAsync(out, c->d_out.p, (size_t)nwin * n.out_dim * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}
