// conv_x3_fp_kernel: bf16x3 implicit GEMM with an LDS-resident input footprint (see the comment below), plus the
// per-shape launcher the cnn_fp_*.hip units instantiate.
#pragma once
#include "conv_common.h"

namespace issk {

// ------------------------------------------------------------------------------------------
// bf16x3 implicit GEMM with an LDS-resident input footprint (kh*kw > 1, Cin % 32 == 0).
//
// conv_x3_kernel re-gathers every A element once per tap: 16 KB of f32 activations per 384 MFMA
// cycles and workgroup, which saturates the per-CU L1/L2 path (~56 B/clk) long before the matrix
// pipe.  Here the k loop is re-ordered to (channel chunk of 32) x (ky, kx): the input pixels a
// 128-row M tile touches form ONE contiguous range [p_lo, p_hi] of the flattened (sample, iy, ix)
// pixel index (NHWC), so per chunk that range is loaded from global memory ONCE, split into
// bf16 hi/lo and kept in LDS; every tap then reads its MFMA A fragments straight from that
// footprint at a per-lane pixel offset (+ ky*W + kx).  Global A traffic drops by ~kh*kw; only
// the 8 KB weight tile per tap still streams (double-buffered) from L2.
constexpr int FPIX = 360;    // footprint capacity in pixels (host-validated per launch): 56 KB of LDS as bf16 hi+lo

// MFMA operand fragments of one k16 step of one tap (A rows hi/lo, two 32-column B tiles hi/lo)
struct Frags { bf16x8 ah, al, b0h, b0l, b1h, b1l; };

// KH x KW are compile-time so that the tap loop unrolls completely: tap coordinates, LDS offsets,
// B-stage parity and the footprint-slice schedule become constants.  (With a run-time tap loop the
// scalar bookkeeping alone was ~80 SALU instructions + a dozen taken branches per 12 MFMAs, more
// than the five issue slots a wave has between two back-to-back MFMAs.)
// Persistent workgroups: a 128-row tile is only 7-14 k cycles of MFMA work, while filling the
// pipeline (footprint + weight loads from HBM/L2, geometry) and draining it (32 stores per lane)
// cost several thousand cycles -- one-tile-per-workgroup launches measured 33 % matrix-pipe
// utilisation.  Here 2 x 256 workgroups each walk a contiguous range of M tiles and the software
// pipeline runs ACROSS tiles: the next tile's first footprint and weight tiles are fetched during
// the current tile's last taps, exactly like the next channel chunk's.
// LDS-DMA: 16 bytes per lane from global memory straight into LDS at (wave-uniform base + lane * 16); no VGPR
// destination, no ds_write.  hipcc does not count this instruction in its s_waitcnt bookkeeping: the kernel
// below waits for it with explicit vmcnt(N) statements.
// Source = wave-uniform base (SGPR pair) + 32-bit unsigned per-lane byte offset: one address VGPR instead of two.
__device__ __forceinline__ void glds16(const void* gbase, unsigned byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(byte_off), "s"(gbase), "s"(lds_dst) : "memory");
}
// The same without saving M0 (two scalar instructions fewer per piece): for kernels in which nothing but these statements uses M0
// (gfx9 DS instructions do not read it; hipcc treats M0 as reserved and re-writes it before any use of its own).
__device__ __forceinline__ void glds16_m0(const void* gbase, unsigned byte_off, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(byte_off), "s"(gbase), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory");
}

// 16-byte chunk swizzle of the unpadded LDS tiles (rows of 4 chunks = 32 bf16): logical chunk c of row r
// lives at physical chunk c ^ ((r >> 2) & 3).  16 lanes of a ds_read_b128 group that read the same logical
// chunk of 16 rows with distinct (r mod 16) hit 16 different 16-byte bank groups.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 4 + (chunk ^ ((row >> 2) & 3)); }

// NH = 2: the workgroup computes TWO 64-column halves (128 output channels) of its M tile from one LDS footprint: the
// tap sequence becomes (tap, half) pairs -- 'virtual taps' -- each with its own 8 KB weight tile, so LDS and the
// pipeline are unchanged while footprint fetch / conversion / LDS writes per MFMA halve for layers with >= 128 channels.
template <int KH, int KW, bool PADDED, bool TR, bool FUSED = false, int NH = 1>
__global__ __launch_bounds__(256, 2) void conv_x3_fp_kernel(const ConvArgs p) {
    constexpr int NT = KH * KW * NH;                 // virtual taps per channel chunk
    static_assert(NH == 1 || NH == 2, "");
    static_assert(NH == 1 || !FUSED, "");
    static_assert(NT >= 2, "1x1 convolutions use conv_x3_kernel");
    constexpr int BSTAGE = BN * 64;                  // bytes of one plane of one weight stage (64 rows x 32 bf16)
    __shared__ __attribute__((aligned(16))) uint16_t sFh[FPIX * 32];
    __shared__ __attribute__((aligned(16))) uint16_t sFl[FPIX * 32];
    __shared__ __attribute__((aligned(1024))) uint16_t sB[4 * 2 * BSTAGE / 2];      // [stage][hi | lo][row][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * BN * NH;
    const int li = lane & 31, lh = lane >> 5;
    const int M = (int)p.M;
    const int totpix = (int)(p.img_stride / p.Cin) * (M / (p.Hq * p.Wq * p.pp));     // samples * H * W

    // contiguous tile range of this workgroup (neighbouring tiles share im2col halos -> same L2 / L1)
    const int per = ((int)p.nblk + (int)gridDim.x - 1) / (int)gridDim.x;
    // Shared first layer: interleaved instead (workgroup i takes tiles i, i + G, i + 2 G, ...) -- all workgroups then walk
    // the same ~70 windows at the same time and the rows of R they share stay in L2 (measured 1.5 % of the step; for
    // ordinary inputs, which are read once, the order makes no difference).
    constexpr bool inter = FUSED;
    const int tstep = inter ? (int)gridDim.x : 1;
    int tile = inter ? (int)blockIdx.x : (int)blockIdx.x * per;
    const int tile_end = inter ? (int)p.nblk : (tile + per < (int)p.nblk ? tile + per : (int)p.nblk);
    if (tile >= tile_end) return;

    // ---- per-tile geometry: first pixel of the tile (uniform) and this lane's A row
    struct Geom { int p_lo, need, lanepix, iy0, ix0; int fy, fx; bool two; };
    // FUSED: per-window scalars of the (at most two) windows a tile's footprint touches.  They are LOADED, and hipcc's
    // s_waitcnt bookkeeping treats a loaded register as possibly pending at every control-flow join: used directly in the
    // tap loop (whose fetch block is conditional) they cost an s_waitcnt vmcnt(0) per tap, which also drains the weight
    // DMAs.  So they are loaded one chunk ahead into `wn`, passed once through `settle` (an opaque asm the compiler must
    // wait in front of) at the start of the tile's last chunk, and only the settled copies are used afterwards.
    struct Win { int wr0, wr1; float mean0, mean1, sd0, sd1; int live0, live1; };
    auto geometry = [&](int t) {
        Geom g;
        const int m0 = t * BM;
        int b, oy, ox;
        map_row32(p, m0, b, oy, ox);
        g.p_lo = (b * p.H + (oy * p.sh - p.pt_)) * p.W + (ox * p.sw - p.pl_);
        {   // pixels the tile really needs (its last row's last tap is the highest, see footprint_fits): slices beyond
            // them are neither converted nor written to LDS -- typically a quarter of the FPIX-pixel buffer
            const int ml = m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1;
            int b2, oy2, ox2;
            map_row32(p, ml, b2, oy2, ox2);
            g.need = (b2 * p.H + (oy2 * p.sh - p.pt_ + KH - 1)) * p.W + (ox2 * p.sw - p.pl_ + KW - 1) - g.p_lo + 1;
        }
        const int m = m0 + wv * 32 + li;
        map_row32(p, m < M ? m : m0, b, oy, ox);
        g.iy0 = oy * p.sh - p.pt_;
        g.ix0 = ox * p.sw - p.pl_;
        int lanepix = (b * p.H + g.iy0) * p.W + g.ix0 - g.p_lo;
        const int hi = FPIX - 1 - ((KH - 1) * p.W + (KW - 1));   // keeps every tap of a row >= M inside the buffer
        g.lanepix = lanepix < 0 ? 0 : (lanepix > hi ? hi : lanepix);
        if (FUSED) {
            // first footprint pixel in the (window, y, x) grid of first-layer outputs (it touches at most two windows:
            // H * W >= FPIX)
            map_row32(p, m0, b, oy, ox);
            g.fy = oy * p.sh; g.fx = ox * p.sw;
            g.two = g.fy * p.W + g.fx + g.need > p.H * p.W;          // footprint reaches into the next window
            const int nb = M / (p.Hq * p.Wq * p.pp);
        }
        return g;
    };
    auto windows_of = [&](int t) {                   // loads only: nothing here may USE the values
        Win w;
        int b, oy, ox;
        map_row32(p, t * BM, b, oy, ox);
        const int nb = M / (p.Hq * p.Wq * p.pp);
        const unsigned b0 = (unsigned)(b < nb ? b : nb - 1), b1 = (unsigned)(b + 1 < nb ? b + 1 : nb - 1);
        w.wr0 = p.win_row[b0]; w.mean0 = p.stats[2u * b0]; w.sd0 = p.stats[2u * b0 + 1u]; w.live0 = p.finite[b0];
        w.wr1 = p.win_row[b1]; w.mean1 = p.stats[2u * b1]; w.sd1 = p.stats[2u * b1 + 1u]; w.live1 = p.finite[b1];
        return w;
    };
    auto settle = [&](Win& w) {
        asm volatile("" : "+v"(w.wr0), "+v"(w.wr1), "+v"(w.mean0), "+v"(w.mean1), "+v"(w.sd0), "+v"(w.sd1), "+v"(w.live0), "+v"(w.live1));
    };
    Geom g = geometry(tile), gn = g;
    Win wc = {}, wn = {}, wx = {};                   // current tile (settled), next tile (pending), target of fetch / stage
    if (FUSED) { wc = windows_of(tile); settle(wc); wx = wc; }

    // ---- weight tiles by LDS-DMA.  Wave wv moves slots [64 wv, 64 wv + 64) of the 256-slot tile: slot = row * 4 +
    // physical chunk; lane reads the LOGICAL chunk that belongs there (swizzle on the source side, LDS linear).
    // Rows >= Cout read row 0 instead: their output columns are never stored.
    const int brow = wv * 16 + (lane >> 2);
    const unsigned boff = 2u * ((unsigned)(n0 + brow < p.Cout ? n0 + brow : 0) * (unsigned)p.Kpad + (unsigned)(((lane & 3) ^ ((brow >> 2) & 3)) * 8));   // bytes
    const unsigned sB_base = (unsigned)(size_t)(&sB[0]);
    auto dma_b = [&](int stage_off, int tap, int c0) {       // stage_off: byte offset of the stage inside sB; tap: virtual
        const unsigned src = boff + 2u * (unsigned)((tap / NH) * p.Cin + c0) + (NH > 1 ? 2u * (unsigned)((tap % NH) * BN * p.Kpad) : 0u);
        glds16(p.wh, src, sB_base + stage_off + wv * 1024);
        glds16(p.wl, src, sB_base + stage_off + BSTAGE + wv * 1024);
    };

    floatx16 acc0, acc1, acc2, acc3;                 // acc2 / acc3: second half (NH = 2 only)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    // ---- footprint prefetch registers: pixel prow + 32 q, channels [c0 + 4 k8, +4).  Pixels outside
    // [0, totpix) are clamped to a valid address: only rows >= M or zero-padded taps ever read them
    // (the latter are zeroed after the LDS read), so the loads need no mask.
    const int k8 = tid & 7, prow = tid >> 3;
    constexpr int NFV = (FPIX + 31) / 32;
    float4 fv[NFV];
    unsigned dbmask = 0;                             // FUSED: bit q = footprint pixel prow + 32 q belongs to the second window
    float4 fsw = make_float4(0.f, 0.f, 0.f, 0.f), fbw = fsw, fps = fsw, fpt = fsw;   // FUSED: weight sums, bias, post-activation
                                                     // scale / shift of this lane's 4 channels
    const int magicW = (65536 + p.W - 1) / p.W;      // x / W == (x * magicW) >> 16 for x < 512, W <= 128
    auto fetch_fp_part = [&](int q, const Geom& gg, int c0) {
        // slices the tile does not need re-load slice 0 (same cache lines; the load COUNT must not change, the vmcnt
        // bookkeeping of the tap loop is exact)
        const int qq = 32 * q < gg.need ? q : 0;
        if (FUSED) {
            int x = gg.fx + prow + 32 * qq;
            const int dy = (x * magicW) >> 16;
            x -= dy * p.W;
            int y = gg.fy + dy;
            const bool second = y >= p.H;
            y -= second ? p.H : 0;
            const int row = y + (second ? wx.wr1 : wx.wr0) - p.f_rmin;
            dbmask = (q == 0 ? 0u : dbmask) | (second ? 1u << q : 0u);       // slice 0 is the first fetch of every chunk
            fv[q] = *reinterpret_cast<const float4*>(p.in + ((unsigned)(row * p.W + x) * (unsigned)p.Cin + (unsigned)(c0 + k8 * 4)));
            return;
        }
        int gp = gg.p_lo + prow + 32 * qq;
        gp = gp < 0 ? 0 : (gp > totpix - 1 ? totpix - 1 : gp);
        fv[q] = *reinterpret_cast<const float4*>(p.in + ((unsigned)gp * (unsigned)p.Cin + (unsigned)(c0 + k8 * 4)));
    };
    auto fetch_chan = [&](int c0) {                  // FUSED: four more loads in the first tap of a chunk (a CONSTANT count:
        const unsigned o = (unsigned)(c0 + k8 * 4);  // without a post-activation affine the bias is loaded three times)
        fsw = *reinterpret_cast<const float4*>(p.f_wsum + o);
        fbw = *reinterpret_cast<const float4*>(p.f_bias + o);
        fps = *reinterpret_cast<const float4*>((p.f_ps ? p.f_ps : p.f_bias) + o);
        fpt = *reinterpret_cast<const float4*>((p.f_pt ? p.f_pt : p.f_bias) + o);
    };
    auto stage_fp = [&](const Geom& gt) {            // gt: geometry of the tile the staged footprint belongs to
        float t0[4], t1[4], rs0 = 0.f, rs1 = 0.f;
        if (FUSED) {
            // scale 1 / std and shift bias - mean / std * sum_k w of (window, channel).  A non-finite window is all zeros
            // in the reference (segmenter.py:86-88): scale 0, shift = bias.
            rs0 = wx.live0 ? 1.0f / wx.sd0 : 0.f;
            rs1 = wx.live1 ? 1.0f / wx.sd1 : 0.f;
            const float mr0 = wx.live0 ? -wx.mean0 * rs0 : 0.f, mr1 = wx.live1 ? -wx.mean1 * rs1 : 0.f;
            const float sw[4] = {fsw.x, fsw.y, fsw.z, fsw.w}, bw[4] = {fbw.x, fbw.y, fbw.z, fbw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { t0[i] = fmaf(sw[i], mr0, bw[i]); t1[i] = fmaf(sw[i], mr1, bw[i]); }
        }
        auto put = [&](int q, float4 v) {
            bf16x4 h, l;
            split4(v, h, l);
            const int o = swz(prow + 32 * q, k8 >> 1) * 8 + (k8 & 1) * 4;    // bf16 units
            *reinterpret_cast<bf16x4*>(&sFh[o]) = h;
            *reinterpret_cast<bf16x4*>(&sFl[o]) = l;
        };
        auto act4 = [&](float4 v) {
            if (p.f_act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (p.f_ps) { v.x = v.x * fps.x + fpt.x; v.y = v.y * fps.y + fpt.y; v.z = v.z * fps.z + fpt.z; v.w = v.w * fps.w + fpt.w; }
            return v;
        };
        // slice q is staged if the tile needs it (uniform) and, for the last slice only, if the lane's pixel exists
        auto live = [&](int q) { return 32 * q < gt.need && (32 * q + 31 < FPIX || prow + 32 * q < FPIX); };
        if (FUSED && !gt.two && p.f_act == 1 && !p.f_ps) {
            // usual case, unswitched by hand (one uniform branch per slice instead of four): the whole footprint lies in
            // one window, relu, no post-activation affine
#pragma unroll
            for (int q = 0; q < NFV; ++q) {
                if (!live(q)) continue;
                const float4 r = fv[q];
                put(q, make_float4(fmaxf(fmaf(r.x, rs0, t0[0]), 0.f), fmaxf(fmaf(r.y, rs0, t0[1]), 0.f),
                                   fmaxf(fmaf(r.z, rs0, t0[2]), 0.f), fmaxf(fmaf(r.w, rs0, t0[3]), 0.f)));
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < NFV; ++q) {
            if (!live(q)) continue;
            float4 v = fv[q];
            if (FUSED) {
                const bool second = (dbmask >> q) & 1u;
                const float sc = second ? rs1 : rs0;
                v = act4(make_float4(fmaf(v.x, sc, second ? t1[0] : t0[0]), fmaf(v.y, sc, second ? t1[1] : t0[1]),
                                     fmaf(v.z, sc, second ? t1[2] : t0[2]), fmaf(v.w, sc, second ? t1[3] : t0[3])));
            }
            put(q, v);
        }
    };

    const int b0s = swz(li, 0), b1s = swz(li + 32, 0);                   // chunk-0 slots of this lane's two weight rows
    const int bx = (li >> 2) & 3;                                        // their swizzle key (same for li and li + 32)
    auto read_frags = [&](Frags& f, const Geom& gg, int ky, int kx, int ks, int stage_off) {
        // opaque copy: without it hipcc hoists the 2 x KH x KW swizzled A addresses (they only depend on the tile)
        // out of the chunk loop and keeps them live in 18-30 VGPRs; recomputing costs ~5 VALU per address
        int lp = gg.lanepix;
        asm volatile("" : "+v"(lp));
        const int pix = lp + ky * p.W + kx;
        const int lc = ks * 2 + lh;
        const int aoff = swz(pix, lc) * 8;                               // bf16 units
        f.ah = *reinterpret_cast<const bf16x8*>(&sFh[aoff]);
        f.al = *reinterpret_cast<const bf16x8*>(&sFl[aoff]);
        if (PADDED) {
            const bool ok = (unsigned)(gg.iy0 + ky) < (unsigned)p.H && (unsigned)(gg.ix0 + kx) < (unsigned)p.W;
            if (!ok) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { f.ah[q] = (__bf16)0.f; f.al[q] = (__bf16)0.f; }
            }
        }
        const int so = stage_off / 2;                                    // bf16 units
        const int c = lc ^ bx;
        f.b0h = *reinterpret_cast<const bf16x8*>(&sB[so + (b0s - (0 ^ bx) + c) * 8]);
        f.b0l = *reinterpret_cast<const bf16x8*>(&sB[so + BSTAGE / 2 + (b0s - (0 ^ bx) + c) * 8]);
        f.b1h = *reinterpret_cast<const bf16x8*>(&sB[so + (b1s - (0 ^ bx) + c) * 8]);
        f.b1l = *reinterpret_cast<const bf16x8*>(&sB[so + BSTAGE / 2 + (b1s - (0 ^ bx) + c) * 8]);
    };
#define ISS_MFMA6(A0, A1)                                                                       \
    if (TR) {                                            /* C^T: rows = channels, columns = pixels (epilogue_tr) */ \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b0h, f.al, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b1h, f.al, A1, 0, 0, 0);                     \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b0l, f.ah, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b1l, f.ah, A1, 0, 0, 0);                     \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b0h, f.ah, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b1h, f.ah, A1, 0, 0, 0);                     \
    } else {                                                                                        \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.b0h, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al, f.b1h, A1, 0, 0, 0);                     \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.b0l, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.b1l, A1, 0, 0, 0);                     \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.b0h, A0, 0, 0, 0);                     \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah, f.b1h, A1, 0, 0, 0);                     \
    }
    auto mfma6 = [&](const Frags& f, int half) {         // half: compile-time after unrolling
        if (NH == 1 || half == 0) { ISS_MFMA6(acc0, acc1) } else { ISS_MFMA6(acc2, acc3) }
    };
#undef ISS_MFMA6

    // ---- software pipeline over the (tile, chunk, tap) sequence.  LDS: three weight stages (B(t), B(t+1), and
    //   the one B(t+2) is being DMA'd into) and the footprint of tap t's chunk.  Tap t:
    //     global: a slice of the next chunk's (or next tile's first) footprint -> registers; DMA B(t+3) -> LDS
    //     LDS   : read the fragments of t's second k16 step          (covered by the 6 MFMAs below)
    //     6 MFMAs on the first k16 step (fragments read during tap t-1)
    //     LDS   : read the fragments of (t+1)'s first k16 step       (covered by the 6 MFMAs below)
    //     6 MFMAs on the second k16 step
    //     wait for this wave's share of B(t+2) (vmcnt counts in order: everything but this tap's own loads), barrier:
    //     the next tap reads B(t+2) in its second half
    //   The last tap of a chunk swaps the footprint between its two MFMA groups (two extra barriers); the last tap
    //   of a tile is followed by the epilogue, whose stores drain behind the next tile's taps.
    constexpr int FPT = (NFV + NT - 2) / (NT - 1);   // footprint slices per tap (taps 0 .. NT-2 of a chunk)
    const int nchunk = p.Cin / XBK;
    if (FUSED) fetch_chan(0);
#pragma unroll
    for (int q = 0; q < NFV; ++q) fetch_fp_part(q, g, 0);
    static_assert(NT >= 3 || NT == 2, "");
    dma_b(0, 0, 0);
    dma_b(2 * BSTAGE, 1 % NT, (1 / NT) * XBK);
    dma_b(4 * BSTAGE, 2 % NT, (2 / NT) * XBK);       // (a 2-tap kernel in a 1-chunk layer re-reads tap 0: harmless, never used)
    stage_fp(g);
    wait_vmcnt<0>();
    __syncthreads();
    Frags fa, fb;                                    // fa: first k16 step of the current tap, fb: second
    read_frags(fa, g, 0, 0, 0, 0);
    int st[4] = {0, 2 * BSTAGE, 4 * BSTAGE, 6 * BSTAGE};   // byte offsets of the stages of B(t), B(t+1), B(t+2), B(t+3) at tap j = 0

    // Wave priorities: the two workgroups of a CU run out of phase; MFMA groups (3) beat the tap's address / load
    // bookkeeping (1-2), and that beats footprint staging and the epilogue (0) of the other workgroup -- 3 % of the step.
    __builtin_amdgcn_s_setprio(2);
    for (; tile < tile_end; tile += tstep) {
        const bool last_tile = tile + tstep >= tile_end;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int c0 = ch * XBK;
            const bool last_chunk = ch + 1 == nchunk;
            // The very last chunk of a workgroup prefetches / stages a footprint and weight tiles nobody will use (from the
            // valid addresses of its last tile) instead of branching around them: with branches in the tap body hipcc's
            // s_waitcnt bookkeeping loses track at every join and puts an s_waitcnt vmcnt(0) in front of each tap's
            // footprint load, which drains the weight DMAs the counted waits below are there to keep in flight.
            constexpr bool fin = false;
            // next tile's geometry one chunk early where possible: the fused variant loads per-window scalars in it
            if (ch == (nchunk >= 2 ? nchunk - 2 : 0) && !last_tile) {
                gn = geometry(tile + tstep);
                if (FUSED) wn = windows_of(tile + tstep);
            }
            if (FUSED) {                             // target windows of this chunk's fetches and of its closing stage_fp
                if (last_chunk && !last_tile) { settle(wn); wx = wn; } else { wx = wc; }
            }
            const Geom gx = last_chunk ? gn : g;                 // tile the next chunk's footprint belongs to (by value:
                                                                 // a runtime choice of references would pin both structs in scratch)
            const int nx_c0 = last_chunk ? 0 : c0 + XBK;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bool has1 = j + 1 < NT || !fin;
                const bool has3 = j + 3 < NT || !fin;
                const int st0 = st[j % 4], st1 = st[(j + 1) % 4], st3 = st[(j + 3) % 4];   // stages of B(t), B(t+1), B(t+3)
                // footprint slices of this tap (compile-time count: the wait below must know it exactly)
                const int q_lo = j < NT - 1 ? (j * FPT < NFV ? j * FPT : NFV) : NFV;
                const int q_hi = j < NT - 1 ? ((j + 1) * FPT < NFV ? (j + 1) * FPT : NFV) : NFV;
                const int nld = (!fin ? q_hi - q_lo : 0) + (FUSED && !fin && j == 0 ? 4 : 0);   // this tap's own loads
                if (!fin) {
                    if (FUSED && j == 0) fetch_chan(nx_c0);
#pragma unroll
                    for (int q = q_lo; q < q_hi; ++q) fetch_fp_part(q, gx, nx_c0);
                }
                if (has3) dma_b(st3, (j + 3) % NT, (j + 3) >= NT ? ((j + 3) >= 2 * NT ? nx_c0 + XBK : nx_c0) : c0);
                read_frags(fb, g, (j / NH) / KW, (j / NH) % KW, 1, st0);
                __builtin_amdgcn_s_setprio(3);
                mfma6(fa, j % NH);
                __builtin_amdgcn_s_setprio(1);
                if (has1) {
                    if (j == NT - 1) {               // tap t+1 opens the next chunk: swap the footprint.  Every wave
                        __syncthreads();             // must have its fb reads back before anyone overwrites it
                        __builtin_amdgcn_s_setprio(0);
                        stage_fp(gx);
                        __builtin_amdgcn_s_setprio(2);
                        __syncthreads();
                    }
                    read_frags(fa, j == NT - 1 ? gx : g, (((j + 1) % NT) / NH) / KW, (((j + 1) % NT) / NH) % KW, 0, st1);
                }
                __builtin_amdgcn_s_setprio(3);
                mfma6(fb, j % NH);
                __builtin_amdgcn_s_setprio(1);
                // B(t+1) was DMA'd during tap t-1; vmcnt counts in order, so allow exactly this tap's own VMEM
                // operations (its footprint loads + 2 DMAs) to stay in flight.  hipcc's own waits for the footprint
                // loads do not know about the DMAs, which only makes them stricter.  (Issuing the loads AFTER the DMAs,
                // which gives them two taps to return, measured 2.4 % slower.)
                if (!has3) wait_vmcnt<0>();
                else if (nld == 0) wait_vmcnt<2>();
                else if (nld == 1) wait_vmcnt<3>();
                else if (nld == 2) wait_vmcnt<4>();
                else if (nld == 3) wait_vmcnt<5>();
                else if (nld == 4) wait_vmcnt<6>();
                else if (nld == 5) wait_vmcnt<7>();
                else if (nld == 6) wait_vmcnt<8>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            // rotate the stage roles by NT taps
            if (NT % 4 != 0) {
                const int t0 = st[0], t1 = st[1], t2 = st[2], t3 = st[3];
                const int r[4] = {t0, t1, t2, t3};
                st[0] = r[NT % 4]; st[1] = r[(NT + 1) % 4]; st[2] = r[(NT + 2) % 4]; st[3] = r[(NT + 3) % 4];
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (TR) {
            epilogue_tr(p, acc0, acc1, (long long)tile * BM + wv * 32 + li, n0, lh);
            if (NH == 2) epilogue_tr(p, acc2, acc3, (long long)tile * BM + wv * 32 + li, n0 + BN, lh);
        } else {
            epilogue_tile(p, acc0, (long long)tile * BM + wv * 32, n0 + li, lh);
            epilogue_tile(p, acc1, (long long)tile * BM + wv * 32, n0 + 32 + li, lh);
            if (NH == 2) {
                epilogue_tile(p, acc2, (long long)tile * BM + wv * 32, n0 + BN + li, lh);
                epilogue_tile(p, acc3, (long long)tile * BM + wv * 32, n0 + BN + 32 + li, lh);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }
        __builtin_amdgcn_s_setprio(2);
        g = gn;
        if (FUSED) wc = wn;
    }
    wait_vmcnt<0>();                                 // the unused prefetches of the last chunk
}

// one filter shape, all (PADDED, TR) variants; fused = shared first layer (ConvArgs::f_*), unpadded shapes with >= 12
// taps only (one footprint slice per tap)
template <int KH, int KW>
void launch_fp_shape(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused, int nh) {
    if constexpr (KH * KW >= 12) {
        if (fused) {
            if (tr) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, true, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, false, true>), grid, dim3(256), 0, st, a);
            return;
        }
    }
    if constexpr (KH == 3 && KW == 3) {              // two 64-column halves per workgroup (see iss_fp_has_nh2)
        if (nh == 2) {
            if (padded && tr) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, true, true, false, 2>), grid, dim3(256), 0, st, a);
            else if (padded) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, true, false, false, 2>), grid, dim3(256), 0, st, a);
            else if (tr) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, true, false, 2>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, false, false, 2>), grid, dim3(256), 0, st, a);
            return;
        }
    }
    if (padded && tr) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, true, true>), grid, dim3(256), 0, st, a);
    else if (padded) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, true, false>), grid, dim3(256), 0, st, a);
    else if (tr) hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_fp_kernel<KH, KW, false, false>), grid, dim3(256), 0, st, a);
}
inline bool iss_fp_has_nh2(int kh, int kw) { return kh == 3 && kw == 3; }

}  // namespace issk

// Filter shapes the footprint kernel is instantiated for (the tap loop is unrolled at compile time); other shapes
// run on conv_x3_kernel.  One extern launcher per shape, defined in the cnn_fp_*.hip units.
#define ISS_FP_SHAPES(X) X(3, 3) X(5, 3) X(3, 5) X(5, 5) X(2, 2) X(4, 4) X(1, 3) X(3, 1)
#define ISS_FP_DECL(KH_, KW_) void iss_fp_launch_##KH_##x##KW_(const issk::ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused, int nh);
ISS_FP_SHAPES(ISS_FP_DECL)
#undef ISS_FP_DECL
#define ISS_FP_DEFINE(KH_, KW_)                                                                               \
    void iss_fp_launch_##KH_##x##KW_(const issk::ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused, int nh) { \
        issk::launch_fp_shape<KH_, KW_>(a, grid, st, padded, tr, fused, nh);                                          \
    }
