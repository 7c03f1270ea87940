// conv_dhl_kernel: the segmenter nets' first dense layer (K = 4992 / 8320 -> 192) as a GEMM whose BOTH operands arrive pre-split, in the
// order the LDS tiles want them -- round 6.
//
// conv_x3_pw_kernel runs that layer at 0.075 of the roofline: a timing-only build whose activation loads all hit L2 is no faster
// (profiles/HISTORY.md, round 6), so it is not memory -- it is the structure: per 32-wide k-tile every thread converts four
// float4 (48 VALU), writes ten LDS vectors and meets a barrier, for 12 MFMAs per wave.  Here nothing is converted:
//   * A: conv_x3_wq3h_kernel<1, true, ..> (conv4) writes its pooled output pixel-major but already split ("PHL": per pixel Cout / 8
//     groups of [hi 8 x 16 bit | lo 8 x 16 bit]; pixels in NHWC order, so a window's pixels x channels are its flattened features and a
//     32-feature k-tile of a window is ONE 128-byte line; the producer's 32 lanes fill such a line with their hi and lo stores).  Eight
//     threads fetch a window's line; in LDS it keeps its eight 16-byte pieces at slot = piece ^ (window & 7) (the thread of
//     (row, slot) loads piece slot ^ (row & 7)), so that the 16 lanes of a fragment read spread over the bank groups;
//   * B: the layer's weights, split into 16-bit halves and packed ONCE per network (dhl_pack_kernel) in the order a k-tile's LDS image
//     wants them -- [k-tile of 32][k-group of 8][hi | lo][column][8 x 16 bit] -- so a k-tile is 24 KB of contiguous memory;
//   * 40 KB k-tiles (16 KB of A + 24 KB of B) in a double buffer, loaded two k-tiles ahead through registers, one barrier per k-tile;
//     128 rows x 192 columns per workgroup (one per CU; a launch of ~30 k windows = 235 tiles), EIGHT waves as 4 x 2 (32 rows x 96
//     columns each: 3 accumulators, 8 fragment reads per 9 MFMAs) -- two waves per SIMD, so that one wave's loads, staging stores and
//     fragment reads run under the other's MFMAs (the one-wave-per-SIMD form ran them one after the other: see the kernel).
// Term and k order per output = conv_x3_pw_kernel's (lo.hi, hi.lo, hi.hi per k16 step, k ascending): bit-identical results.
#pragma once
#include "conv_ws.h"

namespace issk {

constexpr int DHL_BM = 128, DHL_BN = 192, DHL_BK = 32;
constexpr int DHL_A = DHL_BM * DHL_BK * 4;         // 16 KB: 128 rows x 128 B (8 pieces (k-group, part) of 16 B, swizzled by row & 7)
constexpr int DHL_B = DHL_BN * DHL_BK * 4;         // 24 KB: 8 planes x 192 columns x 16 B
constexpr int DHL_STAGE = DHL_A + DHL_B;           // 40 KB

struct DhlArgs {
    const uint16_t* a;       // PHL tensor: [window][K / 8 groups][hi 8 | lo 8] x 16 bit (K * 4 bytes per window)
    const uint16_t* wp;      // packed weights: [column tile of 192][K / 32][4 k-groups][2 parts][192 columns][8]
    const float* bias;       // [Cout] or null
    float* out;              // [M][Cout] f32
    unsigned np;             // windows the tensor has room for (a multiple of 128 >= M: rows beyond M are read, never stored)
    int M, K, Cout, act;     // act: 0 none, 1 relu
};

// NW = 8 waves (two per SIMD; the shipped form) or 4 (one per SIMD: the round's first versions, kept for same-box comparisons behind
// -DISS_DHL_NW=4).  Timing-only builds of the one-wave-per-SIMD form priced a launch as: 36 MFMAs per wave and k-tile alone 0.51 of
// it; + fragment reads and the barrier 0.14; + the staging stores 0.15; + the global loads behind them 0.19 (profiles/HISTORY.md
// round 6) -- three phases of similar size that a single wave runs one after the other, whatever their order.  With two waves per SIMD
// the hardware runs one wave's loads / stores / fragment reads under the other's MFMAs.
#ifndef ISS_DHL_NW
#define ISS_DHL_NW 8
#endif
template <bool F16, int NW = ISS_DHL_NW>
__global__ __launch_bounds__(NW * 64, 1) void conv_dhl_kernel(const DhlArgs p) {
    static_assert(NW == 4 || NW == 8, "");
    constexpr int NT = NW * 64;                                         // threads
    constexpr int RB = 8 / NW;                                          // 32-row blocks per wave: the waves tile 128 x 192 as (NW / 2) x 2
    constexpr int NA = 1024 / NT, NB = 1536 / NT;                       // 16-byte pieces of a k-tile's A / B image per thread
    __shared__ __attribute__((aligned(4096))) unsigned char smem[2 * DHL_STAGE];
    const unsigned s0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wv >> 1, wc = wv & 1;                                // row group of 32 RB rows, 96-column half
    const int m0 = (int)blockIdx.x * DHL_BM;
    const int n0 = (int)blockIdx.y * DHL_BN;                            // column tile (layers wider than 192: the activation is re-read per tile)
    const int nk = p.K / DHL_BK;
    const unsigned rowbytes = (unsigned)p.K * 4u;                       // bytes per window

    // ---- staging: thread t moves A pieces t, t + NT, .. (NA of the tile's 1024: window i >> 3, LDS slot i & 7 <- source piece
    // (i & 7) ^ (window & 7): the swizzle of the fragment reads below) and B pieces t, t + NT, .. (NB of 1536, a straight copy)
    unsigned a_src[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int i = tid + NT * j, row = i >> 3, slot = i & 7;
        a_src[j] = (unsigned)(m0 + row) * rowbytes + (unsigned)((slot ^ (row & 7)) * 16);
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3)))* LdsW16;
    struct Regs { u32x4 a[NA], b[NB]; };
    auto gather = [&](Regs& r, int kt) {
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.a) + (size_t)kt * 128;               // 128 B per window and k-tile
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(p.wp) + ((size_t)blockIdx.y * nk + kt) * DHL_B;
#pragma unroll
        for (int j = 0; j < NA; ++j) r.a[j] = *reinterpret_cast<const u32x4*>(ab + a_src[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) r.b[j] = *reinterpret_cast<const u32x4*>(bb + (unsigned)((tid + NT * j) * 16));
    };
    auto stage = [&](Regs& r, unsigned buf) {
        const unsigned dst = s0 + buf * (unsigned)DHL_STAGE + (unsigned)(tid * 16);
#pragma unroll
        for (int j = 0; j < NA; ++j) *(LdsW16)(dst + (unsigned)(j * NT * 16)) = r.a[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *(LdsW16)(dst + (unsigned)(DHL_A + j * NT * 16)) = r.b[j];
    };

    floatx16 acc[RB][3];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

    // fragment addresses inside a stage.  A: row * 128 + 16 (piece ^ (row & 7)), piece = 2 (2 ks + lh) + part -- the row's base carries
    // (2 lh) ^ (row & 7) and the k16 step / part are XORed in as constants (row & 7 = li & 7: the row blocks start at multiples of 32);
    // B: plane (2 kg + part) at DHL_A + 3072 (2 kg + part), column * 16; k16 step ks of a lane: k-group 2 ks + lh
    const unsigned a_rd = (unsigned)((wr * 32 * RB + li) * 128 + (((2 * lh) ^ (li & 7)) * 16));
    const unsigned b_rd = (unsigned)(DHL_A + (wc * 96 + li) * 16 + lh * 2 * 3072);
    struct Frag { bf16x8 ah[RB], al[RB], bh[3], bl[3]; };
    auto read_frags = [&](Frag& f, unsigned st, int ks) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            f.ah[r] = *(LdsR16)(st + ((a_rd + (unsigned)(r * 32 * 128)) ^ (unsigned)(ks * 4 * 16)));
            f.al[r] = *(LdsR16)(st + ((a_rd + (unsigned)(r * 32 * 128)) ^ (unsigned)(ks * 4 * 16 + 16)));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f.bh[c] = *(LdsR16)(st + b_rd + (unsigned)(ks * 4 * 3072 + c * 32 * 16));
            f.bl[c] = *(LdsR16)(st + b_rd + (unsigned)(ks * 4 * 3072 + 3072 + c * 32 * 16));
        }
    };
    // C^T as conv_x3_pw_kernel computes it (rows = columns of the layer, columns = windows): the same products in the same order per
    // accumulator (lo.hi, hi.lo, hi.hi per k16 step, k ascending); term outermost, so that an accumulator's MFMAs are 3 RB issues apart
    auto mfmas = [&](const Frag& f) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[r][c] = mfma_x3<F16>(term == 1 ? f.bl[c] : f.bh[c], term == 0 ? f.al[r] : f.ah[r], acc[r][c]);
    };

    // two register sets, a double buffer in LDS, one barrier per k-tile: k-tile kt + 2 is requested while kt is multiplied and kt + 1
    // (requested one step earlier) goes into the other buffer
    Regs r0, r1;
    gather(r0, 0);
    gather(r1, nk > 1 ? 1 : 0);
    stage(r0, 0u);
    __syncthreads();
    int kt = 0;
    unsigned cur = 0;
    // timing-only experiment bits (-DISS_DHL_EXP=.., with -DISS_DHL_NW=4 what profiles/r06_dhl_experiments.txt was measured on; wrong
    // results, never in a release build): 4 no staging stores, 8 fragments read once, 16 no global loads, 32 no barrier
#ifndef ISS_DHL_EXP
#define ISS_DHL_EXP 0
#endif
    Frag f0, f1;
    if (ISS_DHL_EXP & 8) { read_frags(f0, s0, 0); read_frags(f1, s0, 1); }
    auto step = [&](Regs& rload, Regs& rstage) {
        const unsigned st = s0 + cur * (unsigned)DHL_STAGE;
        if (!(ISS_DHL_EXP & 8)) read_frags(f0, st, 0);
        if (!(ISS_DHL_EXP & 16)) gather(rload, kt + 2 < nk ? kt + 2 : nk - 1);
        if (!(ISS_DHL_EXP & 8)) read_frags(f1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ISS_DHL_EXP & 4)) stage(rstage, cur ^ 1u);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(ISS_DHL_EXP & 32)) __syncthreads();
        ++kt;
        cur ^= 1u;
    };
    while (true) {
        step(r0, r1);
        if (kt >= nk) break;
        step(r1, r0);
        if (kt >= nk) break;
    }
    // ---- epilogue.  Transposed accumulators: lane li = window (row of the GEMM) m0 + 32 (RB wr + r) + li; register 4 g + i of
    // accumulator c = column wc * 96 + c * 32 + 8 g + 4 lh + i: bias, relu, one float4 per (c, g)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int m = m0 + (wr * RB + r) * 32 + li;
        if (m >= p.M) continue;
        float* orow = p.out + (size_t)m * p.Cout;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wc * 96 + c * 32 + 8 * g + 4 * lh;
                if (n >= p.Cout) continue;
                float4 v = make_float4(acc[r][c][4 * g], acc[r][c][4 * g + 1], acc[r][c][4 * g + 2], acc[r][c][4 * g + 3]);
                if (p.bias) { const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n); v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w; }
                if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(orow + n) = v;
            }
    }
}

// host: the dense row the kernel takes (after a conv_x3_wq3h_kernel<1, ..> launch whose pooled output it reads)
inline bool dhl_supported(int K, int Cout, int act, bool has_ps, bool has_res) {
    return K % DHL_BK == 0 && K >= 2 * DHL_BK && Cout % 4 == 0 && Cout >= 32 && Cout <= 16 * DHL_BN && act <= 1 && !has_ps && !has_res;
}
__host__ __device__ inline int dhl_col_tiles(int Cout) { return (Cout + DHL_BN - 1) / DHL_BN; }
inline unsigned dhl_npad(long long windows) { return (unsigned)((windows + DHL_BM - 1) / DHL_BM * DHL_BM); }
inline size_t dhl_packed_elems(int K, int Cout = DHL_BN) { return (size_t)dhl_col_tiles(Cout) * (K / 8) * 2 * DHL_BN * 8; }   // [column tile][K / 32][4][2][192][8]
void iss_dhl_pack(const uint16_t* wh, const uint16_t* wl, uint16_t* out, int Cout, int Kpad, int K, hipStream_t st);
void iss_dhl_launch(const DhlArgs& a, hipStream_t st, bool f16);

}  // namespace issk
