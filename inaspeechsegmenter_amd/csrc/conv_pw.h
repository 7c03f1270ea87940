// conv_x3_pws_kernel: pointwise (1x1, stride 1, unpadded) convolutions and dense layers as a STREAMING bf16x3 GEMM on the
// NHWC pixel list -- ResNet-101's bottleneck 1x1 layers (resnet.py:48-75; 68 % of the x-vector path's GPU time) and the
// K <= 2048 dense layers of the segmenter networks (segmenter.py:163).  Same arithmetic, tile (128 rows x 64 channels, four
// waves of 32 x 64), LDS operand format and XCD-aware persistent tile order as conv_x3_pw_kernel (cnn.hip), which keeps the
// deep-K first dense layer and is the ISS_NO_PWS=1 fallback.  What is different, and why:
//
// These layers are bound by memory, not by the matrix pipe (K = 32 .. 512: 1 .. 16 k-steps of ~0.2 us per tile, against
// 2 - 4 us of loaded memory latency; the old kernel reached 4.4 TB/s on reads, 2.7 TB/s on writes, 22 % MFMA busy):
//
//  1. A four-deep operand ring in REGISTERS (the register file, 512 KB per CU, is the one buffer large enough to hold the
//     bytes in flight): a k-tile's six loads are issued four k-steps before its MFMAs and converted into LDS three steps
//     later.  hipcc cannot be given that: its waitcnt pass follows flow-graph paths that cannot happen (a tile boundary in
//     every step; a ring set overwritten while "pending") and drains the ring with `s_waitcnt vmcnt(1)` in front of a
//     gather.  So every load of the steady state is issued from inline asm into tied ("+v") register variables, the
//     kernel COUNTS the vector-memory instructions it has issued (`issued`; a mark per ring set and one for the epilogue
//     operands) and waits with `s_waitcnt vmcnt(issued - mark)` rounded DOWN to a literal: vector-memory instructions
//     complete in order, so that is exact when every counted instruction really was issued and conservative otherwise.
//     Counted: the asm loads (unconditional) and the eight unpredicated stores of a full tile.  Not counted (waiting for
//     more than necessary, never for less): predicated stores of edge tiles, anything the compiler issues itself.
//  2. The epilogue goes through a wave-private LDS transpose (32 rows x 36 floats per wave, two 32-channel halves in
//     turn).  In the transposed accumulator layout a lane owns 4 consecutive channels of ONE pixel per register group: a
//     float4 store touches 32 pixel rows with 32 B each -- partial lines, on the store side and on the residual side of
//     every expansion layer.  After the transpose eight lanes cover 128 contiguous bytes of a pixel row: a store or
//     residual-load instruction is 8 full lines.
//  3. The residual and bias of a tile are requested when the PREVIOUS tile's epilogue has consumed its own (the residual
//     no longer depends on the accumulator layout), a whole tile ahead instead of one exposed round trip per tile.
//
// LDS: 60 KB operands + 18 KB epilogue staging = 78 KB, two workgroups per CU.  ~250 VGPRs (RES), no spills.
#pragma once
#include "conv_common.h"

namespace issk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int PWS_NR = 4;                            // ring depth (register sets)
constexpr int PWS_ELD = 36;                          // pitch of the epilogue staging rows (floats): conflict-free ds_write_b128

// s_waitcnt vmcnt(n) for a run-time, wave-uniform n: the largest literal <= n (fewer outstanding = a longer wait = safe).
#define ISS_PWS_W(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ void pws_wait_outstanding(unsigned n) {
    if (n >= 63) { ISS_PWS_W(63); return; }          // the counter has 6 bits; with <= 63 outstanding, an instruction that has 63 younger ones is done
    if (n >= 36) {
        if (n >= 54) ISS_PWS_W(54); else if (n >= 48) ISS_PWS_W(48); else if (n >= 42) ISS_PWS_W(42); else ISS_PWS_W(36);
    } else if (n >= 18) {
        if (n >= 30) ISS_PWS_W(30); else if (n >= 26) ISS_PWS_W(26); else if (n >= 24) ISS_PWS_W(24); else if (n >= 20) ISS_PWS_W(20); else ISS_PWS_W(18);
    } else {
        if (n >= 16) ISS_PWS_W(16); else if (n >= 12) ISS_PWS_W(12); else if (n >= 8) ISS_PWS_W(8); else if (n >= 6) ISS_PWS_W(6); else ISS_PWS_W(0);
    }
}
#define ISS_PWS_LD(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase))

__device__ __forceinline__ void pws_split4(const f32x4 v, bf16x4& h, bf16x4& l) {
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    l[0] = (__bf16)(v[0] - (float)h[0]); l[1] = (__bf16)(v[1] - (float)h[1]);
    l[2] = (__bf16)(v[2] - (float)h[2]); l[3] = (__bf16)(v[3] - (float)h[3]);
}

// RES: residual add (p.res != null).  SIMPLE (host-checked): act <= 1 and no post-activation affine.  Needs p.bias,
// Cout % 4 == 0, pp == 1, Kpad == Cin, byte offsets of one tile below 2^32 (host-checked in cnn.hip).
template <bool RES, bool SIMPLE>
__global__ __launch_bounds__(256, 2) void conv_x3_pws_kernel(const ConvArgs p) {
    constexpr int NR = PWS_NR;
    __shared__ __attribute__((aligned(16))) uint16_t sAh[2][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sAl[2][BM * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBh[2][BN * XLD];
    __shared__ __attribute__((aligned(16))) uint16_t sBl[2][BN * XLD];
    __shared__ __attribute__((aligned(16))) float sE[4 * 32 * PWS_ELD];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ntiles = p.nblk * p.nblk_n;
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    const int k8 = tid & 7, lr = tid >> 3;           // A staging: k columns [4 k8, 4 k8 + 4) of rows lr + 32 j
    const int br = tid >> 2, bseg = tid & 3;         // B staging: 8 bf16 of weight row br, hi and lo
    const int li = lane & 31, lh = lane >> 5;        // MFMA operand / accumulator coordinates
    const int er = lane >> 3, ec = (lane & 7) * 4;   // epilogue: row er + 8 j, 4 channels from ec of a 32 x 32 half tile

    // One tile: uniform 64-bit bases + 32-bit per-lane BYTE offsets (global_load saddr form).  Rows >= M re-read the last
    // row, weight rows / channels >= Cout re-read row / channel 0: their results are never stored.
    struct Tile {
        long long m0; int n0; bool full;
        const float* abase;                          // A rows of the tile
        unsigned ao[4];                              // bytes: row (lr + 32 j), columns from 4 k8
        unsigned wo;                                 // bytes into wh / wl: weight row n0 + br, 8 bf16 from 8 bseg
    };
    auto coords = [&](unsigned tt) {
        Tile T;
        unsigned mt, nt;
        gemm_tile_of_block(tt, p.nblk, p.nblk_n, mt, nt);
        T.m0 = (long long)mt * BM;
        T.n0 = (int)nt * BN;
        T.abase = p.in + T.m0 * p.Cin;
        const int left = (int)(p.M - T.m0 < BM ? p.M - T.m0 : BM);            // rows of this tile that exist
        T.full = left == BM && T.n0 + BN <= p.Cout;
#pragma unroll
        for (int j = 0; j < 4; ++j) T.ao[j] = (unsigned)((lr + 32 * j < left ? lr + 32 * j : left - 1) * p.Cin + k8 * 4) * 4u;
        T.wo = (unsigned)((T.n0 + br < p.Cout ? T.n0 + br : 0) * p.Kpad + bseg * 8) * 2u;
        return T;
    };

    // ---- the ring and the instruction count
    f32x4 ra[NR][4];
    u32x4 rh[NR], rl[NR];
    f32x4 r4[8], b4[2];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        rh[i] = u32x4{0, 0, 0, 0}; rl[i] = u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) r4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    b4[0] = b4[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned issued = 0;                             // vector-memory instructions issued so far that are COUNTED (see the header)
    unsigned mark[NR];                               // `issued` right behind the gather of ring set i
    unsigned mark_epi = 0;                           // ... behind the residual / bias request

    const int nk = p.Kpad / XBK;
    // load cursor (tile, k-tile), NR k-steps ahead of the compute cursor; past the last tile it re-reads that tile
    unsigned tl = t;
    Tile TL = coords(t);
    int ktl = 0;
    Tile TC = TL;                                    // compute cursor
    int ktc = 0;

#define ISS_PWS_GATHER(I)                                                                                            \
    {                                                                                                                \
        const float* sa_ = TL.abase + ktl * XBK;                                                                     \
        const uint16_t* sh_ = p.wh + ktl * XBK;                                                                      \
        const uint16_t* sl_ = p.wl + ktl * XBK;                                                                      \
        ISS_PWS_LD(ra[I][0], TL.ao[0], sa_); ISS_PWS_LD(ra[I][1], TL.ao[1], sa_);                                    \
        ISS_PWS_LD(ra[I][2], TL.ao[2], sa_); ISS_PWS_LD(ra[I][3], TL.ao[3], sa_);                                    \
        ISS_PWS_LD(rh[I], TL.wo, sh_); ISS_PWS_LD(rl[I], TL.wo, sl_);                                                \
        issued += 6; mark[I] = issued;                                                                               \
        if (++ktl == nk) {                                                                                           \
            ktl = 0;                                                                                                 \
            tl = tl + gridDim.x < ntiles ? tl + gridDim.x : tl;                                                      \
            TL = coords(tl);                                                                                         \
        }                                                                                                            \
    }
    // wait for ring set I (everything up to its mark), then convert it into LDS buffer BUF.  The empty asm makes every use
    // of the set depend on a statement that cannot move above the wait.
#define ISS_PWS_STAGE(I, BUF)                                                                                        \
    {                                                                                                                \
        pws_wait_outstanding(issued - mark[I]);                                                                      \
        asm volatile("" : "+v"(ra[I][0]), "+v"(ra[I][1]), "+v"(ra[I][2]), "+v"(ra[I][3]), "+v"(rh[I]), "+v"(rl[I]));     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
            bf16x4 h, l;                                                                                             \
            pws_split4(ra[I][j], h, l);                                                                              \
            *reinterpret_cast<bf16x4*>(&sAh[BUF][(lr + 32 * j) * XLD + k8 * 4]) = h;                                 \
            *reinterpret_cast<bf16x4*>(&sAl[BUF][(lr + 32 * j) * XLD + k8 * 4]) = l;                                 \
        }                                                                                                            \
        *reinterpret_cast<u32x4*>(&sBh[BUF][br * XLD + bseg * 8]) = rh[I];                                           \
        *reinterpret_cast<u32x4*>(&sBl[BUF][br * XLD + bseg * 8]) = rl[I];                                           \
    }

    // residual (in the store layout: 8 full lines per instruction) and bias of tile T
    auto preload_epi = [&](const Tile& T) {
        const int left = (int)(p.M - T.m0 < BM ? p.M - T.m0 : BM);            // rows of this tile that exist (>= 1)
        const float* rbase = p.res + T.m0 * p.Cout;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int c = T.n0 + 32 * t2 + ec;
            const unsigned cs = c < p.Cout ? (unsigned)c : 0u;
            ISS_PWS_LD(b4[t2], cs * 4u, p.bias);
            if (RES) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = wv * 32 + 8 * j + er;
                    ISS_PWS_LD(r4[4 * t2 + j], ((unsigned)((row < left ? row : left - 1) * p.Cout) + cs) * 4u, rbase);
                }
            }
        }
        issued += RES ? 10 : 2;
        mark_epi = issued;
    };

    floatx16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int aoff = (wv * 32 + li) * XLD + lh * 8;
    const int boff_s = li * XLD + lh * 8;
    float* const E = &sE[wv * 32 * PWS_ELD];

    auto epilogue = [&](const Tile& T) {
        pws_wait_outstanding(issued - mark_epi);
        asm volatile("" : "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(r4[5]), "+v"(r4[6]), "+v"(r4[7]),
                          "+v"(b4[0]), "+v"(b4[1]));
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v[0] = t2 == 0 ? acc0[4 * g + 0] : acc1[4 * g + 0];
                v[1] = t2 == 0 ? acc0[4 * g + 1] : acc1[4 * g + 1];
                v[2] = t2 == 0 ? acc0[4 * g + 2] : acc1[4 * g + 2];
                v[3] = t2 == 0 ? acc0[4 * g + 3] : acc1[4 * g + 3];
                *reinterpret_cast<f32x4*>(&E[li * PWS_ELD + 8 * g + 4 * lh]) = v;
            }
            const int c = T.n0 + 32 * t2 + ec;
            const bool cok = c < p.Cout;
            const int cs = cok ? c : 0;
            f32x4 s4 = f32x4{1.f, 1.f, 1.f, 1.f}, t4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!SIMPLE && p.ps) { s4 = *reinterpret_cast<const f32x4*>(p.ps + cs); t4 = *reinterpret_cast<const f32x4*>(p.pt + cs); }
            const long long mrow = T.m0 + wv * 32 + er;
            float* orow = p.out + (size_t)mrow * p.Cout + cs;
            f32x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(8 * j + er) * PWS_ELD + ec]);
                v += b4[t2];
                if (RES) v += r4[4 * t2 + j];
                if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (!SIMPLE && p.act > 1) { v[0] = apply_act(v[0], p.act); v[1] = apply_act(v[1], p.act); v[2] = apply_act(v[2], p.act); v[3] = apply_act(v[3], p.act); }
                if (!SIMPLE && p.ps) v = v * s4 + t4;
                o[j] = v;
            }
            if (T.full) {                                                       // unpredicated: these four stores are counted
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(orow + (size_t)(8 * j) * p.Cout) = o[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (cok && mrow + 8 * j < p.M) *reinterpret_cast<f32x4*>(orow + (size_t)(8 * j) * p.Cout) = o[j];
            }
        }
        if (T.full) issued += 8;
    };

    int cur = 0;
    bool done = false;
    // one k-step: loads for k-tile +NR into ring set LD, MFMAs on LDS buffer `cur`, ring set ST (k-tile +1) into the other buffer
#define ISS_PWS_STEP(LD, ST)                                                                                         \
    {                                                                                                                \
        ISS_PWS_GATHER(LD)                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(2);                                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                           \
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&sAh[cur][aoff + ks * 16]);                           \
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&sAl[cur][aoff + ks * 16]);                           \
            const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(&sBh[cur][boff_s + ks * 16]);                        \
            const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(&sBl[cur][boff_s + ks * 16]);                        \
            const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(&sBh[cur][boff_s + 32 * XLD + ks * 16]);             \
            const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(&sBl[cur][boff_s + 32 * XLD + ks * 16]);             \
            /* C^T: rows = channels, columns = pixels; two independent accumulators alternate */                     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0h, al, acc0, 0, 0, 0);                                  \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1h, al, acc1, 0, 0, 0);                                  \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0l, ah, acc0, 0, 0, 0);                                  \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1l, ah, acc1, 0, 0, 0);                                  \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0h, ah, acc0, 0, 0, 0);                                  \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1h, ah, acc1, 0, 0, 0);                                  \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ISS_PWS_STAGE(ST, cur ^ 1)                                                                                   \
        __syncthreads();                                                                                             \
        cur ^= 1;                                                                                                    \
        if (++ktc == nk) {                           /* tile complete */                                            \
            epilogue(TC);                                                                                            \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }                         \
            ktc = 0;                                                                                                 \
            t += gridDim.x;                                                                                          \
            if (t >= ntiles) done = true;                                                                            \
            else { TC = coords(t); preload_epi(TC); }                                                                \
        }                                                                                                            \
    }

    preload_epi(TC);
    ISS_PWS_GATHER(0) ISS_PWS_GATHER(1) ISS_PWS_GATHER(2) ISS_PWS_GATHER(3)
    ISS_PWS_STAGE(0, 0)
    __syncthreads();
    for (;;) {                                       // every exit is a break straight out of the loop
        ISS_PWS_STEP(0, 1)
        if (done) break;
        ISS_PWS_STEP(1, 2)
        if (done) break;
        ISS_PWS_STEP(2, 3)
        if (done) break;
        ISS_PWS_STEP(3, 0)
        if (done) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (redundant) loads target registers: land before the wave ends
#undef ISS_PWS_STEP
#undef ISS_PWS_STAGE
#undef ISS_PWS_GATHER
}

// ------------------------------------------------------------------------------------------
// conv_x3_pws2_kernel: the residual-free layers with a multiple of 128 output channels (ResNet-101's 512 -> 128 /
// 1024 -> 256 reductions, the transition layers) on 128 x 128 tiles.  With 64-column tiles the two (or four) column tiles
// of a row tile run on neighbouring workgroups of one XCD and are meant to share the A rows through its L2; measured
// (rocprofv3 --pmc FETCH_SIZE per layer, tools/pmc_by_order.py): 512 -> 128 fetches 1.67 x, 512 -> 256 2.10 x, 1024 -> 256
// 1.72 x its algorithmic bytes -- the second column tile misses -- and these layers ran at 3.6 TB/s where the one-column-tile
// layers (256 -> 64, 128 -> 32) reach 5.4.  Here one workgroup computes all 128 columns of its rows: A is read, split and
// staged once per 128 columns.  Same asm loads / counted waits as above with a two-set ring (depth made no difference on
// these read-dominated layers); 24 MFMAs per k-step and wave (four 32 x 32 accumulators).  LDS: 40 KB A + 40 KB W = 80 KB
// exactly (two workgroups per CU), so the epilogue's transpose staging ALIASES the A buffer the last k-step has just
// consumed, with one more barrier per tile before the next k-step's conversion may overwrite it.
// STRIDED: a 1x1 convolution with stride > 1 (the ResNet shortcut projections, resnet.py:60-64) -- the same GEMM on a
// strided pixel list: only the per-tile row addresses change (map_row32 once per tile), the k loop does not.
// DUAL: two A sources (ConvArgs::in2 ...): the k-tiles of `in` (Cin / 32 of them) are followed by those of `in2` on its own,
// possibly strided, pixel list -- a projection shortcut and the expansion it is added to as ONE GEMM over the concatenated K
// (resnet.py:60-75: relu(bn3(conv3(r)) + bn_s(conv_s(x)))): the shortcut's output (as large as the block's) is neither written
// nor read back.  Only the per-tile row addresses and the source of a k-step change; the weights are concatenated at load time.
template <bool SIMPLE, bool STRIDED = false, bool DUAL = false>
__global__ __launch_bounds__(256, 2) void conv_x3_pws2_kernel(const ConvArgs p) {
    static_assert(!(STRIDED && DUAL), "");
    constexpr int BN2 = 128;
    constexpr int ABUF = 2 * BM * XLD * 2;           // bytes of one A buffer: hi | lo
    constexpr int WBUF = 2 * BN2 * XLD * 2;          // bytes of one W buffer: hi | lo
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * ABUF + 2 * WBUF];
    static_assert(2 * ABUF + 2 * WBUF == 81920 && ABUF >= 4 * 32 * PWS_ELD * 4, "LDS budget");
    auto sAh = [&](int b) { return reinterpret_cast<uint16_t*>(smem + b * ABUF); };
    auto sAl = [&](int b) { return reinterpret_cast<uint16_t*>(smem + b * ABUF + ABUF / 2); };
    auto sBh = [&](int b) { return reinterpret_cast<uint16_t*>(smem + 2 * ABUF + b * WBUF); };
    auto sBl = [&](int b) { return reinterpret_cast<uint16_t*>(smem + 2 * ABUF + b * WBUF + WBUF / 2); };

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned ntiles = p.nblk * p.nblk_n;       // nblk_n = Cout / 128 here
    unsigned t = blockIdx.x;
    if (t >= ntiles) return;
    const int k8 = tid & 7, lr = tid >> 3;
    const int br = tid >> 2, bseg = tid & 3;         // W staging: 8 bf16 of weight rows br and br + 64, hi and lo
    const int li = lane & 31, lh = lane >> 5;
    const int er = lane >> 3, ec = (lane & 7) * 4;

    struct Tile { long long m0; int n0; bool full; const float* abase; unsigned ao[4]; unsigned wo[2]; const float* abase2; unsigned ao2[4]; };
    const int nk1 = p.Cin / XBK;                     // DUAL: k-tiles of the first source
    auto coords = [&](unsigned tt) {
        Tile T;
        unsigned mt, nt;
        gemm_tile_of_block(tt, p.nblk, p.nblk_n, mt, nt);
        T.m0 = (long long)mt * BM;
        T.n0 = (int)nt * BN2;
        const int left = (int)(p.M - T.m0 < BM ? p.M - T.m0 : BM);
        T.full = left == BM;                         // Cout % 128 == 0: no column edge
        T.abase2 = nullptr;
        T.ao2[0] = T.ao2[1] = T.ao2[2] = T.ao2[3] = 0u;
        if (DUAL) {                                  // second source: GEMM row m = (b, oy, ox) -> pixel (b, oy * sh2, ox * sw2) of in2
            auto pix2 = [&](int m) {
                int b, oy, ox;
                map_row32(p, m, b, oy, ox);
                return ((long long)b * p.H2 + oy * p.sh2) * p.W2 + ox * p.sw2;
            };
            const long long p0 = pix2((int)T.m0);
            T.abase2 = p.in2 + p0 * p.Cin2;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                T.ao2[j] = (unsigned)((pix2((int)T.m0 + (lr + 32 * j < left ? lr + 32 * j : left - 1)) - p0) * p.Cin2 + k8 * 4) * 4u;
        }
        if (STRIDED) {                               // GEMM row m = output pixel (b, oy, ox) -> input pixel (b, oy * sh, ox * sw)
            auto pix = [&](int m) {
                int b, oy, ox;
                map_row32(p, m, b, oy, ox);
                return ((long long)b * p.H + oy * p.sh) * p.W + ox * p.sw;
            };
            const long long p0 = pix((int)T.m0);
            T.abase = p.in + p0 * p.Cin;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                T.ao[j] = (unsigned)((pix((int)T.m0 + (lr + 32 * j < left ? lr + 32 * j : left - 1)) - p0) * p.Cin + k8 * 4) * 4u;
        } else {
            T.abase = p.in + T.m0 * p.Cin;
#pragma unroll
            for (int j = 0; j < 4; ++j) T.ao[j] = (unsigned)((lr + 32 * j < left ? lr + 32 * j : left - 1) * p.Cin + k8 * 4) * 4u;
        }
        T.wo[0] = (unsigned)((T.n0 + br) * p.Kpad + bseg * 8) * 2u;
        T.wo[1] = (unsigned)((T.n0 + br + 64) * p.Kpad + bseg * 8) * 2u;
        return T;
    };

    f32x4 ra[2][4];
    u32x4 rh[2][2], rl[2][2];
    f32x4 b4[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        rh[i][0] = rh[i][1] = rl[i][0] = rl[i][1] = u32x4{0, 0, 0, 0};
    }
    b4[0] = b4[1] = b4[2] = b4[3] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned issued = 0, mark[2], mark_epi = 0;

    const int nk = p.Kpad / XBK;
    unsigned tl = t;
    Tile TL = coords(t);
    int ktl = 0;
    Tile TC = TL;
    int ktc = 0;

#define ISS_PWS2_GATHER(I)                                                                                           \
    {                                                                                                                \
        const bool s2_ = DUAL && ktl >= nk1;         /* wave-uniform: which source this k-tile comes from */         \
        const float* sa_ = s2_ ? TL.abase2 + (ktl - nk1) * XBK : TL.abase + ktl * XBK;                               \
        const uint16_t* sh_ = p.wh + ktl * XBK;                                                                      \
        const uint16_t* sl_ = p.wl + ktl * XBK;                                                                      \
        const unsigned o0_ = s2_ ? TL.ao2[0] : TL.ao[0], o1_ = s2_ ? TL.ao2[1] : TL.ao[1];                           \
        const unsigned o2_ = s2_ ? TL.ao2[2] : TL.ao[2], o3_ = s2_ ? TL.ao2[3] : TL.ao[3];                           \
        ISS_PWS_LD(ra[I][0], o0_, sa_); ISS_PWS_LD(ra[I][1], o1_, sa_);                                              \
        ISS_PWS_LD(ra[I][2], o2_, sa_); ISS_PWS_LD(ra[I][3], o3_, sa_);                                              \
        ISS_PWS_LD(rh[I][0], TL.wo[0], sh_); ISS_PWS_LD(rl[I][0], TL.wo[0], sl_);                                    \
        ISS_PWS_LD(rh[I][1], TL.wo[1], sh_); ISS_PWS_LD(rl[I][1], TL.wo[1], sl_);                                    \
        issued += 8; mark[I] = issued;                                                                               \
        if (++ktl == nk) {                                                                                           \
            ktl = 0;                                                                                                 \
            tl = tl + gridDim.x < ntiles ? tl + gridDim.x : tl;                                                      \
            TL = coords(tl);                                                                                         \
        }                                                                                                            \
    }
#define ISS_PWS2_STAGE(I, BUF)                                                                                       \
    {                                                                                                                \
        pws_wait_outstanding(issued - mark[I]);                                                                      \
        asm volatile("" : "+v"(ra[I][0]), "+v"(ra[I][1]), "+v"(ra[I][2]), "+v"(ra[I][3]), "+v"(rh[I][0]), "+v"(rl[I][0]),   \
                          "+v"(rh[I][1]), "+v"(rl[I][1]));                                                           \
        uint16_t* ah_ = sAh(BUF); uint16_t* al_ = sAl(BUF); uint16_t* bh_ = sBh(BUF); uint16_t* bl_ = sBl(BUF);      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
            bf16x4 h, l;                                                                                             \
            pws_split4(ra[I][j], h, l);                                                                              \
            *reinterpret_cast<bf16x4*>(&ah_[(lr + 32 * j) * XLD + k8 * 4]) = h;                                      \
            *reinterpret_cast<bf16x4*>(&al_[(lr + 32 * j) * XLD + k8 * 4]) = l;                                      \
        }                                                                                                            \
        *reinterpret_cast<u32x4*>(&bh_[br * XLD + bseg * 8]) = rh[I][0];                                             \
        *reinterpret_cast<u32x4*>(&bl_[br * XLD + bseg * 8]) = rl[I][0];                                             \
        *reinterpret_cast<u32x4*>(&bh_[(br + 64) * XLD + bseg * 8]) = rh[I][1];                                      \
        *reinterpret_cast<u32x4*>(&bl_[(br + 64) * XLD + bseg * 8]) = rl[I][1];                                      \
    }

    auto preload_epi = [&](const Tile& T) {          // bias of the tile's 128 channels, in the store layout
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) ISS_PWS_LD(b4[t2], (unsigned)(T.n0 + 32 * t2 + ec) * 4u, p.bias);
        issued += 4;
        mark_epi = issued;
    };

    floatx16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const int aoff = (wv * 32 + li) * XLD + lh * 8;
    const int boff_s = li * XLD + lh * 8;

    auto epilogue = [&](const Tile& T, int ebuf) {   // ebuf: the A buffer the tile's last k-step has consumed
        float* const E = reinterpret_cast<float*>(smem + ebuf * ABUF) + wv * 32 * PWS_ELD;
        pws_wait_outstanding(issued - mark_epi);
        asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                v[0] = acc[t2][4 * g + 0]; v[1] = acc[t2][4 * g + 1]; v[2] = acc[t2][4 * g + 2]; v[3] = acc[t2][4 * g + 3];
                *reinterpret_cast<f32x4*>(&E[li * PWS_ELD + 8 * g + 4 * lh]) = v;
            }
            const int c = T.n0 + 32 * t2 + ec;
            f32x4 s4 = f32x4{1.f, 1.f, 1.f, 1.f}, t4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!SIMPLE && p.ps) { s4 = *reinterpret_cast<const f32x4*>(p.ps + c); t4 = *reinterpret_cast<const f32x4*>(p.pt + c); }
            const long long mrow = T.m0 + wv * 32 + er;
            float* orow = p.out + (size_t)mrow * p.Cout + c;
            f32x4 o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&E[(8 * j + er) * PWS_ELD + ec]);
                v += b4[t2];
                if (p.act == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                else if (!SIMPLE && p.act > 1) { v[0] = apply_act(v[0], p.act); v[1] = apply_act(v[1], p.act); v[2] = apply_act(v[2], p.act); v[3] = apply_act(v[3], p.act); }
                if (!SIMPLE && p.ps) v = v * s4 + t4;
                o[j] = v;
            }
            if (T.full) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(orow + (size_t)(8 * j) * p.Cout) = o[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (mrow + 8 * j < p.M) *reinterpret_cast<f32x4*>(orow + (size_t)(8 * j) * p.Cout) = o[j];
            }
        }
        if (T.full) issued += 16;
    };

    int cur = 0;
    bool done = false;
#define ISS_PWS2_STEP(LD, ST)                                                                                        \
    {                                                                                                                \
        ISS_PWS2_GATHER(LD)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        __builtin_amdgcn_s_setprio(2);                                                                               \
        {                                                                                                            \
            const uint16_t* ah_ = sAh(cur); const uint16_t* al_ = sAl(cur);                                          \
            const uint16_t* bh_ = sBh(cur); const uint16_t* bl_ = sBl(cur);                                          \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                       \
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&ah_[aoff + ks * 16]);                            \
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(&al_[aoff + ks * 16]);                            \
                bf16x8 bh[4], bl[4];                                                                                 \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                      \
                    bh[c] = *reinterpret_cast<const bf16x8*>(&bh_[boff_s + c * 32 * XLD + ks * 16]);                 \
                    bl[c] = *reinterpret_cast<const bf16x8*>(&bl_[boff_s + c * 32 * XLD + ks * 16]);                 \
                }                                                                                                    \
                /* C^T: rows = channels, columns = pixels; the four accumulators take turns */                       \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[c], al, acc[c], 0, 0, 0); \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[c], ah, acc[c], 0, 0, 0); \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[c], ah, acc[c], 0, 0, 0); \
            }                                                                                                        \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ISS_PWS2_STAGE(ST, cur ^ 1)                                                                                  \
        __syncthreads();                                                                                             \
        cur ^= 1;                                                                                                    \
        if (++ktc == nk) {                           /* tile complete */                                            \
            epilogue(TC, cur ^ 1);                                                                                   \
            _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                            \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;                                      \
            ktc = 0;                                                                                                 \
            t += gridDim.x;                                                                                          \
            if (t >= ntiles) done = true;                                                                            \
            else { TC = coords(t); preload_epi(TC); }                                                                \
            __syncthreads();                         /* the staging aliased an operand buffer */                    \
        }                                                                                                            \
    }

    preload_epi(TC);
    ISS_PWS2_GATHER(0) ISS_PWS2_GATHER(1)
    ISS_PWS2_STAGE(0, 0)
    __syncthreads();
    for (;;) {
        ISS_PWS2_STEP(0, 1)
        if (done) break;
        ISS_PWS2_STEP(1, 0)
        if (done) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ISS_PWS2_STEP
#undef ISS_PWS2_STAGE
#undef ISS_PWS2_GATHER
}

// host: can this launch run on conv_x3_pws_kernel?
inline bool pws_supported(const ConvArgs& a) {
    return a.bias != nullptr && a.Cout % 4 == 0 && a.pp == 1 && a.Kpad == a.Cin && a.Kpad % XBK == 0 &&
           (long long)BM * a.Cin * 4 < (1ll << 31) && (long long)a.Cout * a.Kpad * 2 < (1ll << 31) && (long long)BM * a.Cout * 4 < (1ll << 31);
}
inline bool pws2_supported(const ConvArgs& a) { return pws_supported(a) && !a.res && a.Cout % 128 == 0; }
// strided 1x1: rows of one tile may span several samples; their byte offsets from the tile's first pixel stay below 2^31
inline bool pws2_strided_supported(const ConvArgs& a, int Ho, int Wo) {
    return pws2_supported(a) && a.M < (1ll << 31) && a.act <= 1 && !a.ps &&
           ((long long)(BM / ((long long)Ho * Wo) + 2) * a.img_stride * 4 < (1ll << 31));
}
void iss_pws_launch(const ConvArgs& a, dim3 grid, hipStream_t st);               // cnn_pw.hip
// two-source form (ConvArgs::in2): a = the row that carries the residual, with res cleared, bias / wh / wl / Kpad of the
// concatenated matrix; every byte offset of a tile relative to its first pixel stays below 2^31 in both sources
inline bool pws2_dual_supported(const ConvArgs& a) {
    return a.bias != nullptr && !a.res && !a.ps && a.act <= 1 && a.pp == 1 && a.Cout % 128 == 0 && a.Cin % XBK == 0 && a.Cin2 % XBK == 0 &&
           a.Kpad == a.Cin + a.Cin2 && a.Kpad <= 2048 && a.M < (1ll << 31) && a.sh2 >= 1 && a.sw2 >= 1 &&
           (long long)BM * a.Cin * 4 < (1ll << 31) && (long long)a.Cout * a.Kpad * 2 < (1ll << 31) && (long long)BM * a.Cout * 4 < (1ll << 31) &&
           ((long long)(BM / ((long long)a.Hq * a.Wq) + 2) * a.H2 * a.W2 * a.Cin2 * 4 < (1ll << 31));
}
void iss_pws2_launch(const ConvArgs& a, hipStream_t st, bool strided = false, bool dual = false);   // 128 x 128 tiles: sets its own column tiling and grid

}  // namespace issk
