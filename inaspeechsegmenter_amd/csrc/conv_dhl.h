// conv_dhl_kernel: the segmenter nets' first dense layer (K = 4992 / 8320 -> 192) as a GEMM whose BOTH operands arrive by LDS-DMA
// -- round 6.
//
// conv_x3_pw_kernel runs that layer at 0.075 of the roofline: a timing-only build whose activation loads all hit L2 is no faster
// (profiles/HISTORY.md, round 6), so it is not memory -- it is the structure: per 32-wide k-tile every thread converts four
// float4 (48 VALU), writes ten LDS vectors and meets a barrier, for 12 MFMAs per wave.  Here nothing is converted and nothing is
// staged through registers:
//   * A: conv_x3_wq3h_kernel<1, true, ..> (conv4) writes its pooled output pixel-major but already split ("PHL": per pixel Cout / 8
//     groups of [hi 8 x 16 bit | lo 8 x 16 bit]; pixels in NHWC order, so a window's pixels x channels are its flattened features and a
//     32-feature k-tile of a window is ONE 128-byte line; the producer's 32 lanes fill such a line with their hi and lo stores).  One
//     global_load_lds_dwordx4 fetches the k-tile of EIGHT windows (8 lanes per line); in LDS a window's line keeps its eight 16-byte
//     pieces at slot = piece ^ (window & 7) (swizzle on the SOURCE side: lane (row, slot) asks for piece slot ^ (row & 7)), so that
//     the 16 lanes of a fragment read spread over the bank groups;
//   * B: the layer's weights, split into 16-bit halves and packed ONCE per network (dhl_pack_kernel) in the order a k-tile's LDS image
//     wants them -- [k-tile of 32][k-group of 8][hi | lo][column][8 x 16 bit] -- so a k-tile is 24 KB of contiguous memory;
//   * a three-stage LDS ring of 40 KB k-tiles (16 KB of A + 24 KB of B), two k-tiles in flight, counted vmcnt, one barrier per
//     k-tile = per 36 MFMAs of a wave; 128 rows x 192 columns per workgroup (one per CU; a launch of ~30 k windows = 235 tiles),
//     four waves as 2 x 2 (64 rows x 96 columns each: 6 accumulators, 10 fragment reads per 18 MFMAs).
// Term and k order per output = conv_x3_pw_kernel's (lo.hi, hi.lo, hi.hi per k16 step, k ascending): bit-identical results.
#pragma once
#include "conv_ws.h"

namespace issk {

constexpr int DHL_BM = 128, DHL_BN = 192, DHL_BK = 32;
constexpr int DHL_A = DHL_BM * DHL_BK * 4;         // 16 KB: 128 rows x 128 B (8 pieces (k-group, part) of 16 B, swizzled by row & 7)
constexpr int DHL_B = DHL_BN * DHL_BK * 4;         // 24 KB: 8 planes x 192 columns x 16 B
constexpr int DHL_STAGE = DHL_A + DHL_B;           // 40 KB
constexpr int DHL_NSTAGE = 3;

struct DhlArgs {
    const uint16_t* a;       // PHL tensor: [window][K / 8 groups][hi 8 | lo 8] x 16 bit (K * 4 bytes per window)
    const uint16_t* wp;      // packed weights: [K / 32][4 k-groups][2 parts][192 columns][8]
    const float* bias;       // [Cout] or null
    float* out;              // [M][Cout] f32
    unsigned np;             // windows the tensor has room for (a multiple of 128 >= M: rows beyond M are read, never stored)
    int M, K, Cout, act;     // act: 0 none, 1 relu
};

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv_dhl_kernel(const DhlArgs p) {
    __shared__ __attribute__((aligned(4096))) unsigned char smem[DHL_NSTAGE * DHL_STAGE];
    const unsigned s0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wv >> 1, wc = wv & 1;                                // 64-row half, 96-column half
    const int m0 = (int)blockIdx.x * DHL_BM;
    const int nk = p.K / DHL_BK;
    const unsigned rowbytes = (unsigned)p.K * 4u;                       // bytes per window

    // ---- one k-tile into ring slot `slot`: wave w issues A pieces w, w + 4, .. (16 of 1 KB = 8 windows x 128 B each) and B pieces
    // w, w + 4, .. (24 of 1 KB, contiguous in the packed weights).  A lane of an A piece: window 8 piece + (lane >> 3), LDS slot lane & 7
    // <- source piece (lane & 7) ^ (window & 7)
    const unsigned a_lane = (unsigned)(m0 + (lane >> 3)) * rowbytes + (unsigned)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    auto load_tile = [&](int kt, int slot) {
        const unsigned dst = s0 + (unsigned)(slot * DHL_STAGE);
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.a) + (size_t)kt * 128;               // 128 B per window and k-tile
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = wv + 4 * j;                               // 0..15: windows 8 piece .. 8 piece + 7 of the tile
            glds16_m0(ab, a_lane + (unsigned)(piece * 8) * rowbytes,
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)(piece * 1024))));
        }
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)kt * DHL_B;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int piece = wv + 4 * j;                               // 0..23
            glds16_m0(bb, (unsigned)(piece * 1024 + lane * 16),
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + (unsigned)(DHL_A + piece * 1024))));
        }
    };

    floatx16 acc[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][c][i] = 0.f;

    // fragment addresses inside a stage.  A: row * 128 + 16 (piece ^ (row & 7)), piece = 2 (2 ks + lh) + part -- the row's base carries
    // (2 lh) ^ (row & 7) and the k16 step / part are XORed in as constants (row & 7 = li & 7: the row blocks start at multiples of 32);
    // B: plane (2 kg + part) at DHL_A + 3072 (2 kg + part), column * 16; k16 step ks of a lane: k-group 2 ks + lh
    const unsigned a_rd = (unsigned)((wr * 64 + li) * 128 + (((2 * lh) ^ (li & 7)) * 16));
    const unsigned b_rd = (unsigned)(DHL_A + (wc * 96 + li) * 16 + lh * 2 * 3072);

    load_tile(0, 0);
    if (nk > 1) load_tile(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        const int slot = kt % DHL_NSTAGE;
        // this wave's 10 pieces of k-tile kt have landed (the 10 of kt + 1 may still fly); then every wave's have, and nobody reads
        // the slot k-tile kt + 2 goes into (it held kt - 1) any more
        if (kt + 1 < nk) wait_vmcnt<10>(); else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) load_tile(kt + 2, (kt + 2) % DHL_NSTAGE);
        const unsigned st = s0 + (unsigned)(slot * DHL_STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[2], al[2], bh[3], bl[3];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                ah[r] = *(LdsR16)(st + ((a_rd + (unsigned)(r * 32 * 128)) ^ (unsigned)(ks * 4 * 16)));
                al[r] = *(LdsR16)(st + ((a_rd + (unsigned)(r * 32 * 128)) ^ (unsigned)(ks * 4 * 16 + 16)));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                bh[c] = *(LdsR16)(st + b_rd + (unsigned)(ks * 4 * 3072 + c * 32 * 16));
                bl[c] = *(LdsR16)(st + b_rd + (unsigned)(ks * 4 * 3072 + 3072 + c * 32 * 16));
            }
            // C^T as conv_x3_pw_kernel computes it (rows = columns of the layer, columns = windows): the same products in the same order
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc[r][c] = mfma_x3<F16>(bh[c], al[r], acc[r][c]);
                    acc[r][c] = mfma_x3<F16>(bl[c], ah[r], acc[r][c]);
                    acc[r][c] = mfma_x3<F16>(bh[c], ah[r], acc[r][c]);
                }
        }
    }
    // ---- epilogue.  Transposed accumulators: lane li = window (row of the GEMM) m0 + wr * 64 + r * 32 + li; register 4 g + i of
    // accumulator c = column wc * 96 + c * 32 + 8 g + 4 lh + i: bias, relu, one float4 per (c, g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int m = m0 + wr * 64 + r * 32 + li;
        if (m >= p.M) continue;
        float* orow = p.out + (size_t)m * p.Cout;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = wc * 96 + c * 32 + 8 * g + 4 * lh;
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
    return K % DHL_BK == 0 && K >= 2 * DHL_BK && Cout % 4 == 0 && Cout <= DHL_BN && act <= 1 && !has_ps && !has_res;
}
inline unsigned dhl_npad(long long windows) { return (unsigned)((windows + DHL_BM - 1) / DHL_BM * DHL_BM); }
inline size_t dhl_packed_elems(int K) { return (size_t)(K / 8) * 2 * DHL_BN * 8; }
void iss_dhl_pack(const uint16_t* wh, const uint16_t* wl, uint16_t* out, int Cout, int Kpad, int K, hipStream_t st);
void iss_dhl_launch(const DhlArgs& a, hipStream_t st, bool f16);

}  // namespace issk
