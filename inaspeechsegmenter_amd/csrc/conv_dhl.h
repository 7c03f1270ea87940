// conv_dhl_kernel: the segmenter nets' first dense layer (K = 4992 / 8320 -> 192) as a GEMM whose BOTH operands arrive pre-split, in the
// order the LDS tiles want them -- round 6.
//
// conv_x3_pw_kernel runs that layer at 0.075 of the roofline: a timing-only build whose activation loads all hit L2 is no faster
// (profiles/HISTORY.md, round 6), so it is not memory -- it is the structure: per 32-wide k-tile every thread converts four
// float4 (48 VALU), writes ten LDS vectors and meets a barrier, for 12 MFMAs per wave.  Here nothing is converted and nothing is
// staged through registers:
//   * A: conv_x3_wq3h_kernel<1, true, ..> (conv4) writes its pooled output pixel-major but already split ("PHL": per pixel Cout / 8
//     groups of [hi 8 x 16 bit | lo 8 x 16 bit]; pixels in NHWC order, so a window's pixels x channels are its flattened features and a
//     32-feature k-tile of a window is ONE 128-byte line; the producer's 32 lanes fill such a line with their hi and lo stores).  Eight
//     threads fetch a window's line; in LDS it keeps its eight 16-byte pieces at slot = piece ^ (window & 7) (the thread of
//     (row, slot) loads piece slot ^ (row & 7)), so that the 16 lanes of a fragment read spread over the bank groups;
//   * B: the layer's weights, split into 16-bit halves and packed ONCE per network (dhl_pack_kernel) in the order a k-tile's LDS image
//     wants them -- [k-tile of 32][k-group of 8][hi | lo][column][8 x 16 bit] -- so a k-tile is 24 KB of contiguous memory;
//   * 40 KB k-tiles (16 KB of A + 24 KB of B) in a double buffer, loaded two k-tiles ahead, one barrier per k-tile = per 36 MFMAs of a
//     wave (the first version moved them by LDS-DMA and ran at a CU's DMA fill rate, ~31 GB/s: see the kernel); 128 rows x 192 columns per workgroup (one per CU; a launch of ~30 k windows = 235 tiles),
//     four waves as 2 x 2 (64 rows x 96 columns each: 6 accumulators, 10 fragment reads per 18 MFMAs).
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
    const uint16_t* wp;      // packed weights: [K / 32][4 k-groups][2 parts][192 columns][8]
    const float* bias;       // [Cout] or null
    float* out;              // [M][Cout] f32
    unsigned np;             // windows the tensor has room for (a multiple of 128 >= M: rows beyond M are read, never stored)
    int M, K, Cout, act;     // act: 0 none, 1 relu
};

template <bool F16>
__global__ __launch_bounds__(256, 1) void conv_dhl_kernel(const DhlArgs p) {
    // v2: the first version fetched both operands by LDS-DMA into a three-stage ring and ran at the LDS-DMA fill rate of a CU
    // (40 KB per k-tile at ~31 GB/s = 1.3 us against 0.6 us of MFMAs: MI355X_MICROARCH.md "ldsdma-fill").  The operands are pre-split, so
    // the register path costs no conversion either: ten 16-byte global loads per thread and k-tile, two k-tiles ahead, ten
    // ds_write_b128 into a double buffer -- the load / store path is several times wider than the DMA path.
    __shared__ __attribute__((aligned(4096))) unsigned char smem[2 * DHL_STAGE];
    const unsigned s0 = (unsigned)(size_t)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wv >> 1, wc = wv & 1;                                // 64-row half, 96-column half
    const int m0 = (int)blockIdx.x * DHL_BM;
    const int nk = p.K / DHL_BK;
    const unsigned rowbytes = (unsigned)p.K * 4u;                       // bytes per window

    // ---- staging: thread t moves A pieces t, t + 256, .. (4 of the tile's 1024: window i >> 3, LDS slot i & 7 <- source piece
    // (i & 7) ^ (window & 7): the swizzle of the fragment reads below) and B pieces t, t + 256, .. (6 of 1536, a straight copy)
    unsigned a_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j, row = i >> 3, slot = i & 7;
        a_src[j] = (unsigned)(m0 + row) * rowbytes + (unsigned)((slot ^ (row & 7)) * 16);
#if defined(ISS_DHL_EXP) && (ISS_DHL_EXP & 1)                            // timing-only: every A load from the tile's first 16 KB (cache hits)
        a_src[j] = (unsigned)(blockIdx.x & 63) * 16384u + (unsigned)(i * 16);
#endif
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef u32x4 __attribute__((address_space(3)))* LdsW16;
    struct Regs { u32x4 a[4], b[6]; };
    auto gather = [&](Regs& r, int kt) {
        const unsigned char* ab = reinterpret_cast<const unsigned char*>(p.a) + (size_t)kt * 128;               // 128 B per window and k-tile
        const unsigned char* bb = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)kt * DHL_B;
#if defined(ISS_DHL_EXP) && (ISS_DHL_EXP & 1)
        ab = reinterpret_cast<const unsigned char*>(p.a);
#endif
#if defined(ISS_DHL_EXP) && (ISS_DHL_EXP & 2)                            // timing-only: every B load from the first k-tile
        bb = reinterpret_cast<const unsigned char*>(p.wp);
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) r.a[j] = *reinterpret_cast<const u32x4*>(ab + a_src[j]);
#pragma unroll
        for (int j = 0; j < 6; ++j) r.b[j] = *reinterpret_cast<const u32x4*>(bb + (unsigned)((tid + 256 * j) * 16));
    };
    auto stage = [&](Regs& r, int buf) {
        const unsigned dst = s0 + (unsigned)(buf * DHL_STAGE) + (unsigned)(tid * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) *(LdsW16)(dst + (unsigned)(j * 4096)) = r.a[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) *(LdsW16)(dst + (unsigned)(DHL_A + j * 4096)) = r.b[j];
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

    // One wave per SIMD: nobody else hides an LDS round trip, so the fragments of BOTH k16 steps are requested up front (the second
    // set lands behind the first set's 18 MFMAs) instead of read - wait - use three registers at a time (hipcc's own order for this
    // loop: the matrix pipe 20 % busy).  The staging stores of the next k-tile ride between the two MFMA blocks.
    struct Frag { bf16x8 ah[2], al[2], bh[3], bl[3]; };
    auto read_frags = [&](Frag& f, unsigned st, int ks) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
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
    // accumulator (lo.hi, hi.lo, hi.hi per k16 step); term outermost, so that an accumulator's MFMAs are six issues apart
    auto mfmas = [&](const Frag& f) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[r][c] = mfma_x3<F16>(term == 1 ? f.bl[c] : f.bh[c], term == 0 ? f.al[r] : f.ah[r], acc[r][c]);
    };

    // two register sets: a k-tile's loads are issued two steps before they are written to LDS (past the last k-tile: re-reads it)
    Regs r0, r1;
    gather(r0, 0);
    gather(r1, nk > 1 ? 1 : 0);
    stage(r0, 0);
    __syncthreads();
    int kt = 0;
    auto step = [&](Regs& rload, Regs& rstage, int cur) {               // loads for kt + 2, MFMAs on kt, k-tile kt + 1 into the other buffer
        const unsigned st = s0 + (unsigned)(cur * DHL_STAGE);
        Frag f0, f1;
        read_frags(f0, st, 0);
        __builtin_amdgcn_sched_barrier(0);
        gather(rload, kt + 2 < nk ? kt + 2 : nk - 1);
        read_frags(f1, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f0);
        __builtin_amdgcn_sched_barrier(0);
        stage(rstage, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(f1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        ++kt;
    };
    while (true) {
        step(r0, r1, 0);
        if (kt >= nk) break;
        step(r1, r0, 1);
        if (kt >= nk) break;
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
