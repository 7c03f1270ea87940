// Shared device code of the conv/dense GEMM kernels of libiss_hip.so: argument block, GEMM-row mapping, the fused
// epilogues and the operand loaders.  Included by cnn.hip and by the cnn_fp_*.hip units that instantiate the
// LDS-footprint kernel per filter shape (split so that `make -j` compiles them in parallel).
#pragma once
#include "iss_internal.h"
#include <cmath>
#include <cstring>
#include <algorithm>
#include <cstdlib>
#include <functional>
#include <type_traits>

namespace issk {


typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128;      // GEMM rows (output pixels) per workgroup
constexpr int BN = 64;       // GEMM cols (output channels) per workgroup
constexpr int BK = 16;       // k-tile of the f32 kernel
constexpr int LDK = BK + 4;  // padded LDS row (floats): conflict-free ds_read_b128
constexpr int XBK = 32;      // k-tile of the bf16x3 kernel (two k16 MFMA steps)
constexpr int XLD = XBK + 8; // padded LDS row (bf16): 80 B, conflict-free ds_read_b128
constexpr int KALIGN = 32;   // weight rows are padded to this many k

struct ConvArgs {
    const float* in;
    const float* w;          // [Cout][Kpad] f32
    const uint16_t* wh;      // [Cout][Kpad] bf16 hi part
    const uint16_t* wl;      // [Cout][Kpad] bf16 lo part
    const float* bias;       // [Cout] or null
    const float* ps;         // post-activation scale [Cout] or null
    const float* pt;         // post-activation shift
    const float* res;        // residual, same shape as out, or null
    float* out;
    const int32_t* ktab;     // [Kpad] x {delta, (ky<<16)|kx}
    const int32_t* win_row;  // PATCH mode
    const float* stats;      // PATCH mode: {mean, std} per sample
    const uint8_t* finite;   // PATCH mode
    long long M;             // samples * Hq * Wq * pp   (GEMM rows)
    long long img_stride;    // floats per input sample
    int H, W, Cin, Cout;
    int Hq, Wq;              // output grid the GEMM rows enumerate: pooled grid when pp > 1, else (Ho, Wo)
    int ph, pw, pp;          // fused pool window (1,1,1 = none); rows m = q*pp + (dy*pw + dx)
    int poolkind;            // 0 max, 1 avg
    int H_k, kw;             // kernel height / width (vectorised loaders walk taps)
    int sh, sw, pt_, pl_;
    int row_stride, pix_stride;
    int act, Kpad, mode;
    // ---- shared first layer (conv_x3_fp_kernel<..., FUSED>).  The PATCH conv in front of this conv is linear in its
    // z-normalised window: conv((x - mean_b) / std_b)[c] = (conv(x)[c] - mean_b * sum_k w[c][k]) / std_b.  conv(x) on the RAW
    // log-mel rows is the same for every window that contains the row, so it is computed ONCE per recording row
    // (first_layer_raw_kernel -> `in`, [row - f_rmin][W][Cin], f32) instead of once per window (34 x fewer outputs for
    // 68-row windows every 2 rows), and this kernel applies the per-window affine map, bias and activation while it
    // stages its LDS footprint.  The first layer's per-window output (the largest tensor of the net) never exists.
    const float* f_bias;     // [Cin] first layer bias
    const float* f_wsum;     // [Cin] sum_k w[c][k] of the first layer
    const float* f_ps;       // [Cin] post-activation scale / shift of the first layer (conv -> relu -> BatchNorm), or null
    const float* f_pt;
    int f_act;               // first layer activation: 0 none, 1 relu
    int f_rmin;              // log-mel row of `in`'s first row
    // zero-padded ('same') first layer (conv_x3_ws_kernel<..., FS>): f_wsum is then the table S[W][Cin] of weight sums over the
    // filter columns that see data at column x; the first f_padt / last f_padb rows of window b are rows f_erow0 + b * (f_padt + f_padb) + e
    // of `in` (first_layer_edge_kernel), every other row y of a window whose first log-mel row is wr is row wr + y - f_rmin
    int f_padt, f_padb, f_erow0;
    // exact division of 0 <= n < 2^31 by pp, pw, Hq*Wq, Wq as mulhi + shift (Granlund-Montgomery, N = 31): the footprint
    // kernel decomposes three GEMM rows per tile, and hipcc expands a 32-bit division by a run-time value into ~28 VALU ops
    unsigned dv_mul[4];
    int dv_sh[4];
    unsigned nblk;           // M tiles
    unsigned nblk_n;         // N tiles (generic kernels are launched 1-D: nblk * nblk_n workgroups)
    int dbg;                 // experiment bits of an ISS_EXPERIMENTS build (always 0 in a release build)
    int tmr;                 // conv_x3_wq_kernel: rows per tile (<= 512, multiple of 4; 0 elsewhere)
    // ---- second input of a two-source 1x1 GEMM (conv_x3_pws2_kernel<.., DUAL>): out = act([in | in2 at stride] . [W | W2]^T + b).
    // `in` is the row's own NHWC pixel list (Cin channels); GEMM row (b, oy, ox) reads pixel (b, oy * sh2, ox * sw2) of the
    // (H2, W2, Cin2) images behind `in2`.  wh / wl / Kpad describe the concatenated [Cout][Cin + Cin2] matrix.
    const float* in2;
    int Cin2, H2, W2, sh2, sw2;
    // ---- second GEMM of a chained pair (conv_x3_pwc_kernel): q = act2(x' . W2^T + bias2), x' = this launch's own output
    const uint16_t* wh2;     // [Cout2][Cout] bf16 hi / lo parts of the next row's weights
    const uint16_t* wl2;
    const float* bias2;
    float* out2;
    int act2, Cout2;
    // ---- split-bf16 activation tensors between two footprint kernels (the "CHL" layout, see chl_* below): in_hl / out_hl = the
    // input / output of this launch is one; in_np / out_np = pixels per plane (chl_npad of the tensor's pixel count)
    int in_hl, out_hl;
    unsigned in_np, out_np;
    int f16;                 // operand halves are fp16, not bf16 (ISS_PREC_F16X3): wh / wl point at the fp16 split of the weights, a CHL
                             // input / output holds fp16 planes; only the kernels with an F16 instantiation are launched with it
    int out_f16;             // the shared pooled epilogue (epilogue_impl) writes a CHL output with fp16 (1) or bf16 (0) halves, whatever
                             // the launch's own operand type is
};

// ------------------------------------------------------------------------------------------
// CHL: the activation layout a footprint kernel's producer writes for a footprint kernel that consumes it (round 6).
// An f32 NHWC tensor costs its consumer a global load into registers, ~7 VALU per element for the bf16 hi / lo operand split and two
// LDS stores per float4, once per (tile, 16-channel chunk) -- and a 16-channel chunk of an NHWC pixel is 64 bytes, half a cache
// line, so every line is fetched by two chunk passes.  CHL keeps the SAME 4 bytes per element, already split, laid out the way
// the consumer's LDS footprint wants them: per 16-channel chunk four PLANES
//     plane 4 ch + 2 kh + part : [pixel][8 x bf16]  = channels 16 ch + 8 kh + {0..7} of every pixel; part 0 = hi, 1 = lo
// of `npad` pixels x 16 bytes each.  64 consecutive pixels of a plane are 1 KB of contiguous memory = one LDS-DMA
// wave-instruction (global_load_lds_dwordx4: lane i -> LDS base + 16 i), and they land as 64 consecutive 16-byte MFMA
// A fragments: no register, no VALU, no ds_write, every tap an immediate offset ((ky W + kx) * 16) from one lane address.
// x = hi + lo with hi = bf16_rne(x), lo = bf16_rne(x - hi): the split the consumers do themselves on an f32 input, so the
// MFMA operands -- and therefore the results -- are bit-identical either way (ISS_DIAG_NO_HL keeps f32 between the layers).
// npad = pixel count rounded up to 64 + one footprint (512): a footprint DMA that starts at any valid pixel stays inside its plane.
inline unsigned chl_npad(long long npix) { return (unsigned)((npix + 63) / 64 * 64 + 512); }
inline size_t chl_bytes(long long npix, int C) { return (size_t)chl_npad(npix) * 16u * 4u * (size_t)(C / 16); }
constexpr size_t ISS_ACT_SLACK = 8u << 20;           // bytes every activation buffer has beyond bc * elems * 4 (the padding of a CHL tensor)
inline bool chl_fits(long long npix, int C) { return C % 16 == 0 && chl_bytes(npix, C) <= (size_t)npix * C * 4 + ISS_ACT_SLACK && chl_bytes(npix, C) < 0xFFF00000ull; }

// Host: magic constants of ConvArgs::dv_* for divisor d >= 1 (mul == 0 means d == 1).
inline void set_fast_div(ConvArgs& a, int slot, int d) {
    if (d <= 1) { a.dv_mul[slot] = 0; a.dv_sh[slot] = 0; return; }
    int l = 0;
    while ((1ll << l) < d) ++l;                                  // l = ceil(log2 d) >= 1
    a.dv_mul[slot] = (unsigned)(((1ull << (31 + l)) / (unsigned long long)d) + 1ull);   // < 2^32
    a.dv_sh[slot] = l - 1;
}
template <class P>
__device__ __forceinline__ int fast_div(const P& p, int slot, int n) {
    return p.dv_mul[slot] ? (int)(__umulhi((unsigned)n, p.dv_mul[slot]) >> p.dv_sh[slot]) : n;
}


// GEMM row -> (sample, oy, ox) of the convolution output it stands for
__device__ __forceinline__ void map_row(const ConvArgs& p, long long m, int& b, int& oy, int& ox) {
    long long q = m;
    int dy = 0, dx = 0;
    if (p.pp > 1) {
        q = m / p.pp;
        const int j = (int)(m - q * p.pp);
        dy = j / p.pw; dx = j - dy * p.pw;
    }
    const int hw = p.Hq * p.Wq;
    b = (int)(q / hw);
    const int rem = (int)(q - (long long)b * hw);
    const int qy = rem / p.Wq, qx = rem - qy * p.Wq;
    oy = qy * p.ph + dy;
    ox = qx * p.pw + dx;
}

// 32-bit replica of map_row (the launch guarantees M < 2^31) on host-precomputed reciprocals (ConvArgs::dv_*):
// it runs three times per tile, and run-time integer division is ~28 (32-bit) / ~100 (64-bit) instructions on gfx950.
template <class P>
__device__ __forceinline__ void map_row32(const P& p, int m, int& b, int& oy, int& ox) {
    int q = m, dy = 0, dx = 0;
    if (p.pp > 1) {
        q = fast_div(p, 0, m);                       // m / pp
        const int j = m - q * p.pp;
        dy = fast_div(p, 1, j); dx = j - dy * p.pw;  // j / pw
    }
    const int hw = p.Hq * p.Wq;
    b = fast_div(p, 2, q);                           // q / (Hq * Wq)
    const int rem = q - b * hw;
    const int qy = fast_div(p, 3, rem), qx = rem - qy * p.Wq;    // rem / Wq
    oy = qy * p.ph + dy;
    ox = qx * p.pw + dx;
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed); give every XCD one contiguous
// range of M tiles so that neighbouring tiles (which share im2col halos) hit the same L2.
__device__ __forceinline__ unsigned tile_of_block(unsigned bid, unsigned nblk) {
    const unsigned per = nblk >> 3;
    if (per == 0 || bid >= per * 8) return bid;
    return (bid & 7) * per + (bid >> 3);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}

// ------------------------------------------------------------------------------------------
// Shared epilogue.  C/D layout of the 32x32 MFMAs (f32 and bf16 alike): col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5): every lane holds 4 groups of 4 CONSECUTIVE rows, so a
// fused pool over 2 or 4 consecutive GEMM rows (rows are enumerated pool-window-major, see
// map_row) is a max/mean over registers of one lane -- no shuffles, no LDS.
// ACT: 0 none, 1 relu, -1 = read p.act at run time (sigmoid / tanh); PP: fused pool window size (1, 2, 4);
// HAS_PS: post-activation scale/shift; HAS_RES: residual add.  The common combinations are compiled without any
// per-element branch (the fully generic form, inlined 32 times per tile, was ~6000 ISA lines of mostly skipped code).
// P: ConvArgs, or the EpiArgs subset the weight-stationary kernel loads at the end of a tile group (conv_ws.h)
// One pooled value of channel n at pooled pixel `pix` into a CHL tensor (planes of `np` pixels; see chl_npad below): the operand split
// the consumer would do on an f32 input, two 2-byte stores.  The weight-stationary kernels' pooled epilogue (conv_ws.h) writes it
// for a conv_x3_wq3h_kernel that reads it by LDS-DMA, like conv_x3_wq_kernel's own epilogue does.
__device__ __forceinline__ void chl_store(float* out, unsigned np, int f16, long long pix, int n, float x) {
    uint16_t* o = reinterpret_cast<uint16_t*>(out) + (((size_t)((n >> 4) * 4 + ((n >> 3) & 1) * 2) * np + (size_t)pix) * 8 + (size_t)(n & 7));
    uint16_t h, l;
    if (f16) {
        const _Float16 hh = (_Float16)x, ll = (_Float16)(x - (float)hh);
        h = __builtin_bit_cast(uint16_t, hh); l = __builtin_bit_cast(uint16_t, ll);
    } else {
        const __bf16 hh = (__bf16)x, ll = (__bf16)(x - (float)hh);
        h = __builtin_bit_cast(uint16_t, hh); l = __builtin_bit_cast(uint16_t, ll);
    }
    o[0] = h;
    o[(size_t)np * 8] = l;                          // the lo plane follows the hi plane
}

template <int ACT, int PP, bool HAS_PS, bool HAS_RES, class P>
__device__ __forceinline__ void epilogue_impl(const P& p, const floatx16& acc, long long mrow0, int n, int lh) {
    if (n >= p.Cout) return;
    const float bias = p.bias ? p.bias[n] : 0.f;
    const float s = HAS_PS ? p.ps[n] : 1.f;
    const float sh = HAS_PS ? p.pt[n] : 0.f;
    // max-pool in front of bias + relu where that is exact: x -> fl(x + bias) and relu are monotone, so
    // max_i relu(fl(x_i + bias)) == relu(fl(max_i x_i + bias)) bit for bit -- 4 (2) x fewer VALU operations per pooled output
    if (PP > 1 && (ACT == 0 || ACT == 1) && !HAS_PS && !HAS_RES) {
        if (p.poolkind == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long long mb = mrow0 + 8 * g + 4 * lh;
                if (PP == 4) {
                    if (mb < p.M) {
                        float x = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3])) + bias;
                        if (ACT == 1) x = fmaxf(x, 0.f);
                        if (p.out_np) chl_store(p.out, p.out_np, p.out_f16, mb >> 2, n, x);
                        else p.out[(size_t)(mb >> 2) * p.Cout + n] = x;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i += 2)
                        if (mb + i < p.M) {
                            float x = fmaxf(acc[4 * g + i], acc[4 * g + i + 1]) + bias;
                            if (ACT == 1) x = fmaxf(x, 0.f);
                            if (p.out_np) chl_store(p.out, p.out_np, p.out_f16, (mb + i) >> 1, n, x);
                            else p.out[(size_t)((mb + i) >> 1) * p.Cout + n] = x;
                        }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long long mb = mrow0 + 8 * g + 4 * lh;        // first of this lane's 4 consecutive rows
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float x = acc[4 * g + i] + bias;
            if (HAS_RES) { if (mb + i < p.M) x += p.res[(size_t)(mb + i) * p.Cout + n]; }
            if (ACT == 1) x = fmaxf(x, 0.f);
            else if (ACT == -1) x = apply_act(x, p.act);
            if (HAS_PS) x = x * s + sh;
            v[i] = x;
        }
        if (PP == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (mb + i < p.M) p.out[(size_t)(mb + i) * p.Cout + n] = v[i];
        } else if (PP == 4) {
            if (mb < p.M) {
                const float r = p.poolkind == 0 ? fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))
                                                : (v[0] + v[1] + v[2] + v[3]) * 0.25f;
                p.out[(size_t)(mb >> 2) * p.Cout + n] = r;
            }
        } else {                                             // PP == 2
#pragma unroll
            for (int i = 0; i < 4; i += 2)
                if (mb + i < p.M) {
                    const float r = p.poolkind == 0 ? fmaxf(v[i], v[i + 1]) : (v[i] + v[i + 1]) * 0.5f;
                    p.out[(size_t)((mb + i) >> 1) * p.Cout + n] = r;
                }
        }
    }
}

template <int PP, class P>
__device__ __forceinline__ void epilogue_pp(const P& p, const floatx16& acc, long long mrow0, int n, int lh) {
    const bool ps = p.ps != nullptr;
    if (p.res) {                                             // residual add (ResNet): run-time activation
        if (ps) epilogue_impl<-1, PP, true, true>(p, acc, mrow0, n, lh);
        else epilogue_impl<-1, PP, false, true>(p, acc, mrow0, n, lh);
    } else if (p.act > 1) {                                  // sigmoid / tanh
        if (ps) epilogue_impl<-1, PP, true, false>(p, acc, mrow0, n, lh);
        else epilogue_impl<-1, PP, false, false>(p, acc, mrow0, n, lh);
    } else if (p.act == 1) {
        if (ps) epilogue_impl<1, PP, true, false>(p, acc, mrow0, n, lh);
        else epilogue_impl<1, PP, false, false>(p, acc, mrow0, n, lh);
    } else {
        if (ps) epilogue_impl<0, PP, true, false>(p, acc, mrow0, n, lh);
        else epilogue_impl<0, PP, false, false>(p, acc, mrow0, n, lh);
    }
}

template <class P>
__device__ __forceinline__ void epilogue_tile(const P& p, const floatx16& acc, long long mrow0, int n, int lh) {
    if (p.pp == 1) epilogue_pp<1>(p, acc, mrow0, n, lh);
    else if (p.pp == 4) epilogue_pp<4>(p, acc, mrow0, n, lh);
    else epilogue_pp<2>(p, acc, mrow0, n, lh);
}

// The one combination the segmenter nets' pooled layers use -- relu, non-overlapping max pool over 2 or 4 outputs, no
// post-activation affine, no residual -- as its own entry: epilogue_tile, inlined once per accumulator, carries all 24
// combinations (the weight-stationary kernel's object code was 2.3 MB, seven times its size with this entry alone, and the
// step 3.3 % slower: instruction-cache footprint of code a launch never executes).  Host: epi_is_pool_relu().
template <class P>
__device__ __forceinline__ void epilogue_pool_relu(const P& p, const floatx16& acc, long long mrow0, int n, int lh) {
    if (p.pp == 4) epilogue_impl<1, 4, false, false>(p, acc, mrow0, n, lh);
    else epilogue_impl<1, 2, false, false>(p, acc, mrow0, n, lh);
}
inline bool epi_is_pool_relu(const ConvArgs& a) {
    return a.act == 1 && (a.pp == 2 || a.pp == 4) && a.poolkind == 0 && !a.ps && !a.res;
}
// ... and what the weight-stationary kernel's EPI = 1 forms take: epilogue_pool_relu's own code also carries relu + AVERAGE pool
// (epilogue_impl<1, PP, false, false> falls through to its generic loop when poolkind != 0).  Only the max form may write CHL, and only
// the max form is what the one-wave-per-SIMD kernels' hand-written epilogues compute: they keep asking epi_is_pool_relu.
inline bool epi_is_pool_relu_any(const ConvArgs& a) {
    return a.act == 1 && (a.pp == 2 || a.pp == 4) && (a.poolkind == 0 || a.poolkind == 1) && !a.ps && !a.res;
}
inline bool epi_is_simple_tr(const ConvArgs& a) { return a.act <= 1 && !a.ps && !a.res; }

// Epilogue of TRANSPOSED accumulators (the MFMAs were issued as W-fragment x A-fragment, i.e. C^T): lane = GEMM row
// (pixel) m, register group g of tile t = output channels n0 + 32 t + 8 g + 4 lh + {0..3}.  Every access is a float4:
// bias / scale / shift, the residual, and the store (8 x 16 B per lane and tile pair instead of 32 x 4 B).  Needs
// pp == 1 and Cout % 4 == 0 (parameter offsets in the blob are multiples of 8 floats).
// SIMPLE (host-checked, epi_is_simple_tr): no residual, no sigmoid / tanh, no post-activation affine -- bias and an optional
// relu only.  The generic form inlines expf / tanhf four times per register group; eight calls per kernel made the
// transposed weight-stationary kernels 78-92 KB, more than the 64 KB instruction cache two CUs share.
template <bool PRELOAD_RES = false, class P = ConvArgs, bool SIMPLE = false>
__device__ __forceinline__ void epilogue_tr(const P& p, const floatx16& acc0, const floatx16& acc1, long long m,
                                            int n0, int lh) {
    if (m >= p.M) return;
    float* orow = p.out + (size_t)m * p.Cout;
    const float* rrow = (!SIMPLE && p.res) ? p.res + (size_t)m * p.Cout : nullptr;
    // PRELOAD_RES: all eight residual loads first, back to back: inside the per-group code below each one is waited for
    // on the spot (8 serial round trips to L2 / HBM per tile -- most of the pointwise ResNet layers' time).  Costs 32
    // VGPRs, which the footprint kernels do not have to spare (their residual-free or 3x3 layers gain nothing from it).
    float4 r4[8];
    if (PRELOAD_RES && rrow) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = n0 + 32 * (i >> 2) + 8 * (i & 3) + 4 * lh;
            r4[i] = *reinterpret_cast<const float4*>(rrow + (c < p.Cout ? c : 0));     // columns >= Cout are not stored
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n0 + 32 * t + 8 * g + 4 * lh;
            if (c >= p.Cout) continue;
            float4 v;
            v.x = t == 0 ? acc0[4 * g + 0] : acc1[4 * g + 0];
            v.y = t == 0 ? acc0[4 * g + 1] : acc1[4 * g + 1];
            v.z = t == 0 ? acc0[4 * g + 2] : acc1[4 * g + 2];
            v.w = t == 0 ? acc0[4 * g + 3] : acc1[4 * g + 3];
            if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + c); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
            if (rrow) {
                const float4 r = PRELOAD_RES ? r4[4 * t + g] : *reinterpret_cast<const float4*>(rrow + c);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
            }
            if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (!SIMPLE && p.act > 1) { v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act); }
            if (!SIMPLE && p.ps) {
                const float4 s4 = *reinterpret_cast<const float4*>(p.ps + c), t4 = *reinterpret_cast<const float4*>(p.pt + c);
                v.x = v.x * s4.x + t4.x; v.y = v.y * s4.y + t4.y; v.z = v.z * s4.z + t4.z; v.w = v.w * s4.w + t4.w;
            }
#ifdef ISS_EXPERIMENTS                                               // store-tail experiments (make EXPERIMENTS=1): never in a release build
            if constexpr (std::is_same<P, ConvArgs>::value) {
                if (p.dbg & 1) {                                     // nontemporal stores
                    __builtin_nontemporal_store(v.x, orow + c); __builtin_nontemporal_store(v.y, orow + c + 1);
                    __builtin_nontemporal_store(v.z, orow + c + 2); __builtin_nontemporal_store(v.w, orow + c + 3);
                    continue;
                }
                if ((p.dbg & 2) && g != 0) continue;                 // a quarter of the stores (wrong results: timing only)
            }
#endif
            *reinterpret_cast<float4*>(orow + c) = v;
        }
    }
}

// Unconditional loads: a load inside a divergent `if` makes hipcc put an `s_waitcnt vmcnt(0)` at the
// join, right behind the load, which exposes the full memory latency in every k iteration.  So:
// always load from a valid address (the tensor base when the element is out of bounds) and
// select afterwards.
__device__ __forceinline__ float4 ld4_or_zero(const float* base, long long off, bool ok) {
    const float4 v = *reinterpret_cast<const float4*>(base + (ok ? off : 0));
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ uint4 ldu4_or_zero(const uint16_t* base, size_t off, bool ok) {
    const uint4 v = *reinterpret_cast<const uint4*>(base + (ok ? off : 0));
    return ok ? v : make_uint4(0, 0, 0, 0);
}

// per-thread gather bookkeeping of one A row
struct RowSrc {
    long long base;
    int iy0, ix0;
    float mean, sd;
    bool ok;
    int b;                   // MODE 4: the window (sample) of the row, for its edge rows
};

template <int MODE>
__device__ __forceinline__ RowSrc row_source(const ConvArgs& p, long long m) {
    RowSrc r;
    r.ok = m < p.M;
    int b, oy, ox;
    map_row(p, r.ok ? m : 0, b, oy, ox);
    r.iy0 = oy * p.sh - p.pt_;
    r.ix0 = ox * p.sw - p.pl_;
    r.mean = 0.f; r.sd = 1.f;
    r.b = b;
    if (MODE == 3 || MODE == 4) {
        // shared first layer read by the generic kernel: `in` = the first conv on the RAW log-mel rows, once per row; GEMM row's sample
        // b is a window whose first row is win_row[b].  mean / sd carry the window's affine map: value = relu(R * sd + (bias - mean' ...))
        // with sd := 1 / std and mean := -mean / std (0 / 0 for a non-finite window: its outputs are replaced by the caller)
        // (MODE 4, zero-padded first layer: base = the window's first shared row only -- the row of a tap is decided per tap, see gather)
        r.base = MODE == 4 ? (long long)(p.win_row[b] - p.f_rmin)
                           : ((long long)(p.win_row[b] - p.f_rmin) + r.iy0) * p.row_stride + (long long)r.ix0 * p.pix_stride;
        const bool live = p.finite[b] != 0;
        const float rstd = live ? 1.0f / p.stats[2 * b + 1] : 0.f;
        r.sd = rstd;
        r.mean = live ? -p.stats[2 * b] * rstd : 0.f;
    } else if (MODE == 2) {
        r.base = (long long)p.win_row[b] * 24 + (long long)r.iy0 * 24 + r.ix0;
        r.mean = p.stats[2 * b];
        r.sd = p.stats[2 * b + 1];
        r.ok = r.ok && p.finite[b];
    } else if (MODE == 1 && p.win_row) {             // window of the resident vbx features: frame (start + ix), feature iy
        r.base = ((long long)p.win_row[b] + r.ix0) * p.pix_stride + (long long)r.iy0 * p.row_stride;
    } else {
        r.base = (long long)b * p.img_stride + (long long)r.iy0 * p.row_stride + (long long)r.ix0 * p.pix_stride;
    }
    return r;
}

// one A element through the im2col table (scalar path: any Cin, and the z-normalised PATCH input)
template <int MODE>
__device__ __forceinline__ float gather_scalar(const ConvArgs& p, const RowSrc& r, int k) {
    const int2 e = reinterpret_cast<const int2*>(p.ktab)[k];
    const int iy = r.iy0 + (e.y >> 16), ix = r.ix0 + (e.y & 0xffff);
    const bool ok = r.ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    float x = p.in[ok ? r.base + e.x : 0];
    if (MODE == 2) x = (x - r.mean) / r.sd;
    return ok ? x : 0.f;
}

// 1-D launch of an (M tiles x N tiles) GEMM grid with the N tiles of one M tile placed on ONE XCD, back to back:
// workgroup id runs on XCD id % 8 (observed), so XCD x takes the M tiles congruent to x (mod 8) and walks their N tiles
// fastest.  The A tile of an M tile is then fetched from HBM once and re-read from that XCD's L2 by the other N tiles
// (with a 2-D grid the 8 N tiles of a 512-channel layer landed on 8 different XCDs and re-fetched A 8 times).
__device__ __forceinline__ void gemm_tile_of_block(unsigned bid, unsigned nblk_m, unsigned nblk_n, unsigned& mtile, unsigned& ntile) {
    const unsigned groups = nblk_m >> 3;                  // complete groups of 8 M tiles
    const unsigned cut = groups * 8 * nblk_n;             // workgroups covered by the XCD-aware mapping
    if (bid < cut) {
        const unsigned x = bid & 7, l = bid >> 3;
        mtile = x + 8 * (l / nblk_n);
        ntile = l % nblk_n;
    } else {                                              // the last < 8 M tiles: plain order
        const unsigned r = bid - cut;
        mtile = groups * 8 + r / nblk_n;
        ntile = r % nblk_n;
    }
}

// ------------------------------------------------------------------------------------------
// Operand halves of the split arithmetic, by 16-bit type (round 6: ISS_PREC_F16X3).  x = hi + lo with hi = rne16(x),
// lo = rne16(x - hi): bf16 keeps 8 + 8 mantissa bits (|x - hi - lo| <= 2^-17 |x|, the dropped lo x lo term 2^-16), fp16
// 11 + 11 (2^-23 / 2^-22) at the same three MFMAs per k-step -- tests/precision_emulation.py: 10-15 x less error in the
// log-probabilities -- for operands inside fp16's range (|x| < 65504; the lo part of a small x is a SUBNORMAL fp16, which
// v_mfma_f32_32x32x16_f16 honours exactly: tools/microbench/mfma_f16_denorm.hip).  The 16-bit pairs travel in `unsigned`s and
// the fragments in bf16x8 registers whatever the type: only the conversions and the MFMA opcode differ.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <bool F16>
__device__ __forceinline__ floatx16 mfma_x3(const bf16x8& a, const bf16x8& b, const floatx16& c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// (lo 16 bits = rne16(a), hi 16 bits = rne16(b)) -- one instruction either way on gfx950
template <bool F16>
__device__ __forceinline__ unsigned cvt_pk16(float a, float b) {
    unsigned r;
    if constexpr (F16) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool F16>
__device__ __forceinline__ float unpk16_lo(unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    else return __uint_as_float(u << 16);
}
template <bool F16>
__device__ __forceinline__ float unpk16_hi(unsigned u) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    else return __uint_as_float(u & 0xffff0000u);
}
// split of four values into (h01, h23, l01, l23)
template <bool F16>
__device__ __forceinline__ void split4_pk(const float4 v, unsigned& h01, unsigned& h23, unsigned& l01, unsigned& l23) {
    h01 = cvt_pk16<F16>(v.x, v.y); h23 = cvt_pk16<F16>(v.z, v.w);
    l01 = cvt_pk16<F16>(v.x - unpk16_lo<F16>(h01), v.y - unpk16_hi<F16>(h01));
    l23 = cvt_pk16<F16>(v.z - unpk16_lo<F16>(h23), v.w - unpk16_hi<F16>(h23));
}

// x = hi + lo with hi = bf16(x), lo = bf16(x - hi)  (v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ void split4(const float4 v, bf16x4& h, bf16x4& l) {
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
    l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
}

}  // namespace issk
