// conv_x3_wq_kernel: weight-stationary bf16x3 implicit GEMM, ONE wave per SIMD, two LDS footprints, every non-MFMA
// instruction placed in the shadow of an MFMA.
//
// The shared-first-layer convolution (conv2 of both segmenter nets: 5x3, 64 -> 64, fused 2x2 max-pool) is 60 % of a step.
// conv_x3_ws_kernel (conv_ws.h) runs it with 8 waves x (64 rows x 64 channels), two waves per SIMD, ONE footprint: the next
// footprint waits, converted, in registers and is written between two barriers at every block boundary, the accumulators are
// drained in a serial epilogue, and with two waves per SIMD every VALU / LDS instruction of one wave comes out of the other
// wave's MFMA issue: 62 % matrix-pipe busy (profiles/r03_pmc_closing.md).
//
// Here: 256 threads = 4 waves, one per SIMD, 512 registers per lane (256 accumulator registers + 256 VGPRs), each wave
// 128 rows x 64 channels of a tile of <= 512 rows, two tiles per group.  Timing-only builds of the first version of this
// kernel (profiles/HISTORY.md, round 4) showed that a wave alone on its SIMD runs its MFMAs at 93 % of the pipe when nothing
// else is issued, and that every other instruction costs its issue time UNLESS it sits directly behind an MFMA with at most
// ~4 companions (a 32-cycle MFMA hides ~5 single-issue instructions, MI355X_MICROARCH.md) -- two MFMAs issued back to back
// waste the first one's shadow.  So:
//   * the stream is 24 slots per tap, ONE MFMA each, with at most ~4 filler instructions behind it (sched_barrier pins);
//   * footprint layout with IMMEDIATE tap offsets: 64 bytes per pixel = [k-half a: hi 16 B | lo 16 B][k-half b: hi | lo],
//     the two 32-byte halves swapped on odd rows (a one-bit swizzle that keeps the two rows of a 2 x 2 pool window on
//     different banks).  A lane keeps two running addresses (even / odd filter rows): hi at +64 kx, lo at +64 kx + 16 --
//     one v_add per filter row instead of six VALU per tap and row block;
//   * TWO footprints of 800 pixels beside the 60 KB of resident weights = exactly 160 KB: tile t of a group lives in
//     footprint t; the next block's footprint is fetched, converted and stored into the OTHER buffer in pieces of <= 6
//     instructions spread over the slots of the current block;
//   * ONE barrier per block, at the start of its last tap (every fragment read of the block has been issued and waited
//     for, every store into the other footprint is done): behind it the next block's first fragments are read and, every
//     second block, the next 16-channel chunk's weights are fetched by LDS-DMA, under the last tap's MFMAs;
//   * NO serial epilogue: tile 0's accumulators are complete one block before tile 1's, so they are pooled / biased /
//     stored in pieces behind the MFMAs of the group's last block, tile 1's behind the first block of the NEXT group, whose
//     first MFMAs take C = 0 instead of zeroed registers (five copies of the block body: plain x 2, zero-C, zero-C +
//     epilogue of the previous tile 1, epilogue of tile 0; ~45 KB of code, inside the 64 KB instruction cache).
// Tiles have `tmr` <= 512 rows (host: the largest multiple of 4 whose footprint fits 800 pixels; 508 for 17-column inputs):
// rows >= tmr of a wave's last row block are computed and dropped.
//
// Compiled for the one combination the dominant launch uses: first-layer-fused, unpadded, relu + 2 x 2 max-pool epilogue.
#pragma once
#include "conv_ws.h"

namespace issk {

constexpr int WQ_ROW = 64;                         // bytes per footprint pixel
constexpr int WQ_PIX = 800;                        // footprint capacity in pixels (host-validated per launch)
constexpr int WQ_FB = WQ_PIX * WQ_ROW;             // 51 200 bytes per footprint
constexpr int WQ_NFV = (WQ_PIX + 63) / 64;         // 64-pixel slices per footprint: 256 threads x 4 channels each (the last one half used)
constexpr int wq_lds_bytes(int nt) { return nt * F2_BST + 2 * WQ_FB; }

// ISS_WQ_EXP (timing-only experiment builds, wrong results; never defined in a release build): bit 0 = no footprint
// pipeline, bit 1 = no barriers / weight refresh, bit 2 = no epilogue stores, bit 3 = no fragment reads, bit 4 = A reads from one
// address for all lanes (no bank conflicts), bit 5 = 8-byte A reads, bit 6 = no wait / barrier at the chunk boundary
#ifndef ISS_WQ_EXP
#define ISS_WQ_EXP 0
#endif

// OUT_HL (round 6): the pooled output is written in the CHL layout (conv_common.h) for a footprint kernel that reads it by LDS-DMA
// (conv_wq3h.h): per pooled value the operand split (4 VALU) and two 2-byte stores instead of one 4-byte store.
// F16 (round 6): fp16 instead of bf16 operand halves (conv_common.h mfma_x3 / cvt_pk16): the same instruction count.
template <int KH, int KW, bool OUT_HL = false, bool F16 = false>
__global__ __launch_bounds__(256, 1) void conv_x3_wq_kernel(const ConvArgs p) {
    constexpr bool X_NOPIPE = ISS_WQ_EXP & 1, X_NOBAR = ISS_WQ_EXP & 2, X_NOEPI = ISS_WQ_EXP & 4, X_NOREAD = ISS_WQ_EXP & 8;
    constexpr bool X_BCAST = ISS_WQ_EXP & 16, X_HALF = ISS_WQ_EXP & 32, X_NOCHUNKBAR = ISS_WQ_EXP & 64;
    constexpr int NT = KH * KW;
    constexpr int G = 2;
    static_assert(KH == 5 && KW == 3 && WQ_NFV == 13, "the slot schedule below is written for 15 taps and 13 slices");
    static_assert(wq_lds_bytes(NT) <= 160 * 1024 && (NT * F2_BST) % 4096 == 0 && WQ_FB % 2048 == 0, "");
    __shared__ __attribute__((aligned(4096))) unsigned char smem[wq_lds_bytes(NT)];     // [NT weight tiles][footprint 0][footprint 1]
    const unsigned sB_base = (unsigned)(size_t)smem;
    const unsigned sF0 = sB_base + NT * F2_BST;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3
    const int n0 = blockIdx.y * BN;
    const int li = lane & 31, lh = lane >> 5;
    const int M = (int)p.M;
    const int TMR = p.tmr;                           // rows per tile (<= 512, multiple of 4)
    const int ntiles = (M + TMR - 1) / TMR;
    const int ngroups = (ntiles + G - 1) / G;
    int grp = (int)blockIdx.x;
    if (grp >= ngroups) return;

    // ---- geometry (parameters through the kernel-argument pointer, reciprocals from the host)
    auto geo_args = [&]() {
        KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        GeoArgs ga;
        ga.H = q->H; ga.W = q->W; ga.Hq = q->Hq; ga.Wq = q->Wq; ga.ph = q->ph; ga.pw = q->pw; ga.pp = q->pp;
        ga.sh = q->sh; ga.sw = q->sw; ga.pt_ = q->pt_; ga.pl_ = q->pl_;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ga.dv_mul[i] = q->dv_mul[i]; ga.dv_sh[i] = q->dv_sh[i]; }
        return ga;
    };
    struct TGeo { int p_lo, need, fy, fx, wb; };     // uniform per tile: first pixel, pixels needed, (y, x) of it, its window
    auto clamp_tile = [&](int t) { return t < ntiles ? t : ntiles - 1; };
    auto geo_uniform = [&](const GeoArgs& ga, int tile) {
        TGeo u;
        const int m0 = clamp_tile(tile) * TMR;
        int b, oy, ox;
        map_row32(ga, m0, b, oy, ox);
        u.p_lo = (b * ga.H + oy) * ga.W + ox;
        u.fy = oy; u.fx = ox; u.wb = b;
        const int ml = m0 + TMR - 1 < M - 1 ? m0 + TMR - 1 : M - 1;
        int b2, oy2, ox2;
        map_row32(ga, ml, b2, oy2, ox2);
        u.need = (b2 * ga.H + (oy2 + KH - 1)) * ga.W + (ox2 + KW - 1) - u.p_lo + 1;
        return u;
    };
    // LDS byte address of the lane's first tap in footprint `fb` for EVEN filter rows: pixel base + the 32-byte half that holds
    // this lane's k-half on that pixel's row (half = k-half ^ row parity); odd filter rows use this address ^ 32
    auto geo_lane = [&](const GeoArgs& ga, int tile, int rb, const TGeo& u, int fb) {
        const int m0 = clamp_tile(tile) * TMR;
        const int m = m0 + (wv * 4 + rb) * 32 + li;
        int b, oy, ox;
        map_row32(ga, m < M ? m : m0, b, oy, ox);
        const int lp = (b * ga.H + oy) * ga.W + ox - u.p_lo;
        const int hi = WQ_PIX - 1 - ((KH - 1) * ga.W + (KW - 1));       // keeps every tap of a row >= M (or >= tmr) inside the buffer
        return sF0 + (unsigned)(fb * WQ_FB) + (unsigned)((lp < 0 ? 0 : (lp > hi ? hi : lp)) * WQ_ROW) + (unsigned)(((lh ^ oy) & 1) << 5);
    };
    struct Win { int wr0, wr1; float mean0, mean1, sd0, sd1; int live0, live1; };
    int nwin;
    { const int spp = p.Hq * p.Wq * p.pp; nwin = M / spp; }
    auto windows_of = [&](int b) {                   // loads only: nothing here may USE the values (see conv_fp.h)
        Win w;
        const unsigned b0 = (unsigned)(b < nwin ? b : nwin - 1), b1 = (unsigned)(b + 1 < nwin ? b + 1 : nwin - 1);
        w.wr0 = p.win_row[b0]; w.mean0 = p.stats[2u * b0]; w.sd0 = p.stats[2u * b0 + 1u]; w.live0 = p.finite[b0];
        w.wr1 = p.win_row[b1]; w.mean1 = p.stats[2u * b1]; w.sd1 = p.stats[2u * b1 + 1u]; w.live1 = p.finite[b1];
        return w;
    };
    auto settle = [&](const Win& w) {
        Win s;
        s.wr0 = __builtin_amdgcn_readfirstlane(w.wr0); s.wr1 = __builtin_amdgcn_readfirstlane(w.wr1);
        s.mean0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.mean0)));
        s.mean1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.mean1)));
        s.sd0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.sd0)));
        s.sd1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(w.sd1)));
        s.live0 = __builtin_amdgcn_readfirstlane(w.live0); s.live1 = __builtin_amdgcn_readfirstlane(w.live1);
        return s;
    };

    // ---- weights of one 16-channel chunk: NT tiles of 4 KB = 4 NT pieces of 1 KB; wave w moves pieces w, w + 4, ...:
    // piece i has (half, plane) = (i & 1, (i >> 1) & 1) = (w & 1, w >> 1) for every one of a wave's pieces, tap = i >> 2.
    // Slot permutation on the source side as in conv_ws.h (conflict-free B reads).
    const int w_half = wv & 1, w_plane = (wv >> 1) & 1;
    unsigned boff_w;
    {
        const int n = 32 * w_half + (lane >> 1), h = (lane & 1) ^ ((n >> 3) & 1);
        const int row = n0 + n;
        boff_w = 2u * ((unsigned)(row < p.Cout ? row : 0) * (unsigned)p.Kpad + (unsigned)(h * 8));      // bytes
    }
    unsigned wdst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_base + w_plane * 2048 + w_half * 1024));     // the wave's piece in tap 0's tile
    auto load_weight_tap = [&](int c0, int v) {      // v: compile-time
        const uint16_t* src = (w_plane ? p.wl : p.wh) + (v * p.Cin + c0);
        asm volatile("" : "+s"(wdst0));              // (one scalar add per piece instead of NT hoisted destinations)
        glds16(src, boff_w, wdst0 + (unsigned)(v * F2_BST));
    };
    unsigned bread = sB_base + (unsigned)((2 * li + (lh ^ ((li >> 3) & 1))) * 16);
    asm volatile("" : "+v"(bread));                  // opaque base: the per-tap offsets stay immediates

    // ---- footprint slices: thread -> pixel 64 q + (tid >> 2), channels [c0 + 4 (tid & 3), + 4) = k-half cg >> 1, 8-byte half cg & 1
    const int cg = tid & 3, prow = tid >> 2;
    float4 fv[WQ_NFV];
    unsigned fm[WQ_NFV];                             // per slice: bit 0 = pixel belongs to the second window, bit 1 = it lies on an odd row
    float4 fsw = make_float4(0.f, 0.f, 0.f, 0.f), fbw = fsw;     // weight sums / bias of this thread's 4 first-layer channels
    Win wx = {}, wpend = {};
    const int magicW = (65536 + p.W - 1) / p.W;      // x / W == (x * magicW) >> 16 (host-checked range)
    const unsigned cin4 = (unsigned)p.Cin * 4u, cg16 = (unsigned)cg * 16u;
    auto fetch_consts = [&](int c0) {
        const unsigned o = (unsigned)(c0 + cg * 4);
        fsw = *reinterpret_cast<const float4*>(p.f_wsum + o);
        fbw = *reinterpret_cast<const float4*>(p.f_bias + o);
    };
    // fetch of slice q in three pieces: (1) pixel -> (dy, x); (2) window, row, masks; (3) address + load
    int f_x, f_dy, f_row;
    auto fetch_1 = [&](const TGeo& u, int q) {       // q: compile-time
        const int qq = 64 * q < u.need ? q : 0;      // unneeded slices re-load slice 0
        int pr = prow;
        if (q == WQ_NFV - 1 && WQ_PIX % 64 != 0) pr = prow < WQ_PIX % 64 ? prow : 0;     // (pixels beyond the buffer: never written)
        int x = u.fy * p.W + u.fx + pr + 64 * qq;
        asm volatile("" : "+v"(x));                // (opaque: the same tile is fetched once per chunk, and hipcc would keep all of it)
        f_dy = (x * magicW) >> 16;
        f_x = x - f_dy * p.W;
    };
    auto fetch_2 = [&](int q) {
        int y = f_dy;
        const bool second = y >= p.H;
        y -= second ? p.H : 0;
        fm[q] = (second ? 1u : 0u) | ((unsigned)(y & 1) << 1);
        f_row = y + (second ? wx.wr1 : wx.wr0) - p.f_rmin;
    };
    auto fetch_3 = [&](int c0, int q) {
        fv[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.in + c0) + ((unsigned)(f_row * p.W + f_x) * cin4 + cg16));
    };
    const float f_lob = p.f_act == 1 ? 0.f : -INFINITY;
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f}, rs0 = 0.f, rs1 = 0.f;
    float mr0 = 0.f, mr1 = 0.f;
    auto conv_consts_1 = [&]() {
        asm volatile("" : "+v"(fsw.x), "+v"(fsw.y), "+v"(fsw.z), "+v"(fsw.w), "+v"(fbw.x), "+v"(fbw.y), "+v"(fbw.z), "+v"(fbw.w));
        rs0 = wx.live0 ? 1.0f / wx.sd0 : 0.f;
        rs1 = wx.live1 ? 1.0f / wx.sd1 : 0.f;
        mr0 = wx.live0 ? -wx.mean0 * rs0 : 0.f; mr1 = wx.live1 ? -wx.mean1 * rs1 : 0.f;
    };
    auto conv_consts_2 = [&]() {
        t0[0] = fmaf(fsw.x, mr0, fbw.x); t0[1] = fmaf(fsw.y, mr0, fbw.y); t0[2] = fmaf(fsw.z, mr0, fbw.z); t0[3] = fmaf(fsw.w, mr0, fbw.w);
    };
    auto conv_consts_3 = [&]() {
        t1[0] = fmaf(fsw.x, mr1, fbw.x); t1[1] = fmaf(fsw.y, mr1, fbw.y); t1[2] = fmaf(fsw.z, mr1, fbw.z); t1[3] = fmaf(fsw.w, mr1, fbw.w);
    };
    // thread's store offset inside a footprint for EVEN rows: pixel, 32-byte half = its k-half, 8-byte half; odd rows: ^ 32
    const unsigned wofs = (unsigned)(prow * WQ_ROW + (cg >> 1) * 32 + (cg & 1) * 8);
    const int wsgn = (cg >> 1) ? -32 : 32;           // ^ 32 on an address whose bit 5 is (cg >> 1), as an add
    // conversion of slice q in six pieces of <= 6 instructions; two register sets (steps 12 and 13 convert two slices each)
    typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
    struct Cv { float sc, ta, tb, tc, td; float4 v; bf16x4 h; };
    auto convert_1 = [&](Cv& c, int q) {             // select the window's scale / shifts
        const bool second = fm[q] & 1u;
        c.sc = second ? rs1 : rs0;
        c.ta = second ? t1[0] : t0[0]; c.tb = second ? t1[1] : t0[1]; c.tc = second ? t1[2] : t0[2]; c.td = second ? t1[3] : t0[3];
    };
    auto convert_2 = [&](Cv& c, int q) {             // affine map
        float4 v = fv[q];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // (the load is waited for here, not earlier)
        c.v = make_float4(fmaf(v.x, c.sc, c.ta), fmaf(v.y, c.sc, c.tb), fmaf(v.z, c.sc, c.tc), fmaf(v.w, c.sc, c.td));
    };
    auto convert_3 = [&](Cv& c) {                    // activation
        c.v.x = fmaxf(c.v.x, f_lob); c.v.y = fmaxf(c.v.y, f_lob); c.v.z = fmaxf(c.v.z, f_lob); c.v.w = fmaxf(c.v.w, f_lob);
    };
    auto convert_4 = [&](Cv& c) {                    // hi parts
        if constexpr (F16) { u32x2h hh; hh[0] = cvt_pk16<true>(c.v.x, c.v.y); hh[1] = cvt_pk16<true>(c.v.z, c.v.w); c.h = __builtin_bit_cast(bf16x4, hh); }
        else { c.h[0] = (__bf16)c.v.x; c.h[1] = (__bf16)c.v.y; c.h[2] = (__bf16)c.v.z; c.h[3] = (__bf16)c.v.w; }
    };
    auto convert_5 = [&](Cv& c) {                    // residuals
        if constexpr (F16) {
            const u32x2h hh = __builtin_bit_cast(u32x2h, c.h);
            c.v = make_float4(c.v.x - unpk16_lo<true>(hh[0]), c.v.y - unpk16_hi<true>(hh[0]), c.v.z - unpk16_lo<true>(hh[1]), c.v.w - unpk16_hi<true>(hh[1]));
        } else c.v = make_float4(c.v.x - (float)c.h[0], c.v.y - (float)c.h[1], c.v.z - (float)c.h[2], c.v.w - (float)c.h[3]);
    };
    auto convert_6 = [&](Cv& c, int q, unsigned wbase) {      // lo parts + the two 8-byte stores; wbase = footprint base + wofs
        bf16x4 l;
        if constexpr (F16) {
            u32x2h ll;
            ll[0] = cvt_pk16<true>(c.v.x, c.v.y); ll[1] = cvt_pk16<true>(c.v.z, c.v.w);
            l = __builtin_bit_cast(bf16x4, ll);
        } else {
        l[0] = (__bf16)c.v.x; l[1] = (__bf16)c.v.y; l[2] = (__bf16)c.v.z; l[3] = (__bf16)c.v.w;
        }
        const unsigned a = wbase + (unsigned)(wsgn * (int)((fm[q] >> 1) & 1u));
        // the last slice is half a slice (32 pixels = the threads of waves 0 and 1): a wave-uniform (scalar) branch
        static_assert(WQ_PIX % 64 == 0 || WQ_PIX % 64 == 32, "");
        if (q == WQ_NFV - 1 && WQ_PIX % 64 != 0 && wv >= 2) return;
        *(LdsW8)(a + (unsigned)(q * 64 * WQ_ROW)) = c.h;
        *(LdsW8)(a + (unsigned)(q * 64 * WQ_ROW + 16)) = l;
    };

    // ---- fragments (double-buffered by tap parity)
    struct AFr { bf16x8 h, l; };
    struct BHi { bf16x8 h0, h1; };                   // hi plane of the weights: double-buffered by tap parity (used from slot 0 on)
    struct BLo { bf16x8 l0, l1; };                   // lo plane: ONE set, read in the first slots of the step that uses it from slot 16
                                                     // on; the last tap's (read in front of the block's barrier) has a set of its own
    const unsigned wstep2 = (unsigned)(2 * p.W * WQ_ROW);       // two filter rows down
    auto mfma = [&](const bf16x8& a, const bf16x8& b, const floatx16& c) {
        return mfma_x3<F16>(a, b, c);
    };

    // accumulators: acc<tile><row block><column block>, named (arrays passed by reference end up in scratch)
    floatx16 c000, c001, c010, c011, c020, c021, c030, c031, c100, c101, c110, c111, c120, c121, c130, c131;
    {
        floatx16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        c000 = z; c001 = z; c010 = z; c011 = z; c020 = z; c021 = z; c030 = z; c031 = z;
        c100 = z; c101 = z; c110 = z; c111 = z; c120 = z; c121 = z; c130 = z; c131 = z;
    }

    // ---- epilogue pieces: accumulator (rb, cb) of a tile, register group g (4 consecutive rows = one pool window):
    // max over the window, + bias, relu, one dword store.  ebase = byte offset of (pooled row of the wave's first row + lh,
    // column n0 + li) in `out`; tile_rows = rows of the tile that exist (the tmr / M bounds).
    struct Epi { const float* bias; float* out; int cout; };
    Epi ep;
    {
        KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        ep.bias = q->bias; ep.out = q->out; ep.cout = q->Cout;
    }
    float ebias0 = 0.f, ebias1 = 0.f;                // bias of columns n0 + li and n0 + 32 + li (one launch = one layer)
    if (ep.bias) { ebias0 = ep.bias[n0 + li < ep.cout ? n0 + li : 0]; ebias1 = ep.bias[n0 + 32 + li < ep.cout ? n0 + 32 + li : 0]; }
    const bool ecol0 = n0 + li < ep.cout, ecol1 = n0 + 32 + li < ep.cout;
    // Stores go through a buffer descriptor over `out`: an offset beyond its size is dropped by the hardware, so rows beyond
    // the tile (tmr) or the launch (M) and columns >= Cout need a select on the offset, not a branch.
    const unsigned hl_np16 = OUT_HL ? p.out_np * 16u : 0u;      // bytes per CHL plane
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, OUT_HL ? (int)(p.out_np * (unsigned)ep.cout * 4u)
                                                                                              : (int)((unsigned)(M >> 2) * (unsigned)ep.cout * 4u), 0x00020000);
    constexpr unsigned E_INVALID = 0xFFFF0000u;      // + the largest scalar offset below (< 16 KB) stays below 2^32: no wrap-around;
                                                     // the host keeps the output below 0xFFF00000 bytes
    const int wrow = wv * 128 + 4 * lh;              // first row of the wave's pool windows of register group 0 inside the tile
    int rowb = ep.cout * 4;                          // bytes per pooled output row
    float e_x;
    unsigned e_off;
    auto epi_a = [&](const floatx16& acc, int g) {   // window maximum, first half
        e_x = fmaxf(acc[4 * g], acc[4 * g + 1]);
        asm volatile("" : "+v"(e_x));
    };
    auto epi_b = [&](const floatx16& acc, int cb, int g) {      // second half, + bias, relu
        e_x = fmaxf(fmaxf(e_x, acc[4 * g + 2]), acc[4 * g + 3]);
        e_x = fmaxf(e_x + (cb ? ebias1 : ebias0), 0.f);
        asm volatile("" : "+v"(e_x));
    };
    // vb0 / vb1: byte offset of (the wave's first pooled row + lh, column n0 [+ 32] + li) of the tile, or E_INVALID for a
    // column >= Cout; tile_rows: rows of the tile that exist
    // OUT_HL pieces.  hipcc's fmaxf canonicalises each accumulator register first (two extra v_max per window) and rebuilds the
    // row bound per unit (two v_mov): with the window maximum as v_max3 + v_max, the bound as one per-block register and the
    // split's conversions written out, a CHL unit costs 12 VALU + 2 stores where the f32 unit costs 10 + 1
    unsigned e_h = 0, e_l = 0;                       // bf16 bits (low half) of the value's hi / lo parts
    int e_lim = 0;                                   // rows of the tile that exist - the lane's first row: unit (rb, g) is stored iff rb * 32 + 8 g < e_lim
    unsigned e_inv = E_INVALID;
    auto epih_a = [&](const floatx16& acc, int g) {
        asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(e_x) : "v"(acc[4 * g]), "v"(acc[4 * g + 1]), "v"(acc[4 * g + 2]));
    };
    auto epih_b = [&](const floatx16& acc, int cb, int g) {
        asm volatile("v_max_f32 %0, %0, %1" : "+v"(e_x) : "v"(acc[4 * g + 3]));
        e_x = fmaxf(e_x + (cb ? ebias1 : ebias0), 0.f);
        asm volatile("" : "+v"(e_x));
    };
    auto epi_s = [&]() {                             // x = hi + lo, as the consumers split an f32 input (conv_common.h split4)
        e_h = cvt_pk16<F16>(e_x, e_x);
        const float r = e_x - unpk16_lo<F16>(e_h);
        e_l = cvt_pk16<F16>(r, r);
    };
    auto epih_c = [&](int rb, int cb, int g, unsigned vb) {
        // (the unit's pixel offset rides in the instruction's immediate field, the plane in the scalar offset: E_INVALID + 480 is
        // still beyond the tensor)
        unsigned sel;                                // (rb * 32 + 8 g < e_lim) ? vb : E_INVALID in two instructions (a literal beside vcc would
                                                     // be a second constant-bus operand: the invalid offset waits in a register, e_inv)
        asm volatile("v_cmp_lt_i32 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %1, vcc" : "=v"(sel) : "v"(vb), "n"(rb * 32 + 8 * g), "v"(e_lim), "v"(e_inv) : "vcc");
        const unsigned vo = sel + (unsigned)((rb * 8 + 2 * g) * 16);
        if (X_NOEPI) return;
        unsigned np16 = hl_np16;
        asm volatile("" : "+s"(np16));
        // channel n0 + 32 cb + li = chunk (n0 >> 4) + 2 cb + (li >> 4): planes 8 cb further on; lo part: the next plane
        __builtin_amdgcn_raw_buffer_store_b16((short)e_h, orsrc, (int)vo, cb ? (int)(np16 << 3) : 0, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)e_l, orsrc, (int)vo, cb ? (int)((np16 << 3) + np16) : (int)np16, 0);
    };
    auto epi_c = [&](int rb, int cb, int g, unsigned vb0, unsigned vb1, int tile_rows) {
        int wr = wrow;
        asm volatile("" : "+v"(wr), "+s"(rowb));     // (keeps hipcc from hoisting 32 row indices / 32 scalar offsets out of the loop)
        const bool ok = wr < tile_rows - (rb * 32 + 8 * g);
        e_off = ok ? (cb ? vb1 : vb0) : E_INVALID;
        if (X_NOEPI) return;
        if (!OUT_HL)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e_x), orsrc, (int)e_off, (rb * 8 + 2 * g) * rowb, 0);
    };
    // byte offset in `out` of (pooled row (tile * tmr + wave's first row) / 4 + lh, column n0 + li); < 2^32 (host-checked)
    auto epi_base = [&](int tile, int cb) {
        const bool col = cb ? ecol1 : ecol0;
        if (OUT_HL)      // CHL byte offset of (pooled pixel tile * tmr / 4 + wv * 32 + lh, channel n0 + li) -- hi part; the host guarantees Cout % 64 == 0
            return (unsigned)(((unsigned)(((n0 >> 4) + (li >> 4)) * 4 + ((li >> 3) & 1) * 2) * p.out_np + (unsigned)(tile * (TMR >> 2) + wv * 32 + lh)) * 16u + (unsigned)((li & 7) * 2));
        return col ? (unsigned)(((tile * (TMR >> 2) + wv * 32 + lh) * ep.cout + n0 + cb * 32 + li) * 4) : E_INVALID;
    };
    auto tile_rows_of = [&](int tile) { const int r = M - tile * TMR; return tile < ntiles ? (r < TMR ? r : TMR) : 0; };

    // ---- prologue: geometry of the first group; its first footprint converted serially into footprint 0
    TGeo ug[G + 2];                                  // the group's tiles + the next group's first two tiles
    unsigned lb[G][4];                               // lane bases (even filter rows) of the group's tiles (tile t reads footprint t)
    unsigned lbn[4];                                 // lane bases of the NEXT group's tile 0 (valid during a group's last chunk)
    auto group_geometry = [&](int g0) {
        const GeoArgs ga = geo_args();
#pragma unroll
        for (int t = 0; t < G; ++t) {
            ug[t] = geo_uniform(ga, g0 * G + t);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) lb[t][rb] = geo_lane(ga, g0 * G + t, rb, ug[t], t);
        }
    };
    group_geometry(grp);
    ug[G] = ug[0]; ug[G + 1] = ug[1];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) lbn[rb] = lb[0][rb];
    wx = settle(windows_of(ug[0].wb)); wpend = windows_of(ug[1].wb);      // (blocks reload wpend in their step 13)
    fetch_consts(0);
#pragma unroll
    for (int q = 0; q < WQ_NFV; ++q) { fetch_1(ug[0], q); fetch_2(q); fetch_3(0, q); }
    conv_consts_1(); conv_consts_2(); conv_consts_3();
    Cv cva, cvb;
#pragma unroll
    for (int q = 0; q < WQ_NFV; ++q) {
        convert_1(cva, q); convert_2(cva, q); convert_3(cva); convert_4(cva); convert_5(cva); convert_6(cva, q, sF0 + wofs);
    }
#pragma unroll
    for (int v = 0; v < NT; ++v) load_weight_tap(0, v);
    wait_vmcnt<0>();
    __syncthreads();

    AFr a[2][4];
    BHi bh[2];
    BLo bl, blast;
    unsigned re[4], ro[4];                           // running addresses: even / odd filter rows of the current tile
    // tap (ky, kx) of a row block: hi at (running address of ky's parity) + 64 kx, lo at + 64 kx + 16 (immediates)
    typedef const bf16x4 __attribute__((address_space(3)))* LdsR8;
    auto read_a_h = [&](AFr& f, unsigned ad, int kx) {
        if (X_BCAST) ad = sF0;
        if (X_HALF) { const bf16x4 t = *(LdsR8)(ad + (unsigned)(kx * WQ_ROW)); f.h[0] = t[0]; f.h[1] = t[1]; f.h[2] = t[2]; f.h[3] = t[3]; }
        else f.h = *(LdsR16)(ad + (unsigned)(kx * WQ_ROW));
    };
    auto read_a_l = [&](AFr& f, unsigned ad, int kx) {
        if (X_BCAST) ad = sF0;
        if (X_HALF) { const bf16x4 t = *(LdsR8)(ad + (unsigned)(kx * WQ_ROW + 16)); f.l[0] = t[0]; f.l[1] = t[1]; f.l[2] = t[2]; f.l[3] = t[3]; }
        else f.l = *(LdsR16)(ad + (unsigned)(kx * WQ_ROW + 16));
    };
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        re[rb] = lb[0][rb]; ro[rb] = (lb[0][rb] ^ 32u) + (unsigned)(p.W * WQ_ROW);
        read_a_h(a[0][rb], re[rb], 0); read_a_l(a[0][rb], re[rb], 0);
    }
    bh[0].h0 = *(LdsR16)(bread); bh[0].h1 = *(LdsR16)(bread + 1024);

    const int nchunk = p.Cin / F2_CH;                // >= 2 (host-checked)
    const int gstep = (int)gridDim.x;
    int prev_tile1 = ntiles;                         // tile whose accumulators (set 1) wait for their epilogue; none yet
    for (; grp < ngroups; grp += gstep) {
        const bool last_group = grp + gstep >= ngroups;
        // one block = one tile x one chunk.  Compile time: t (tile = footprint = fragment-set parity of tap 0), ZC (first chunk:
        // the first MFMA of every accumulator takes C = 0), EP (epilogue pieces of the OTHER accumulator set ride along)
        auto run_block = [&](const int t, const bool ZC, const bool EP, const int c0, const bool last_chunk,
                             floatx16& d00, floatx16& d01, floatx16& d10, floatx16& d11,
                             floatx16& d20, floatx16& d21, floatx16& d30, floatx16& d31,
                             const floatx16& o00, const floatx16& o01, const floatx16& o10, const floatx16& o11,
                             const floatx16& o20, const floatx16& o21, const floatx16& o30, const floatx16& o31,
                             const int etile) __attribute__((always_inline)) {
            // the footprint this block builds (for the block after it) and the one after that (whose windows it loads)
            const TGeo un = t == 0 ? ug[1] : (last_chunk ? ug[G] : ug[0]);
            const TGeo un2 = t == 0 ? (last_chunk ? ug[G] : ug[0]) : (last_chunk ? ug[G + 1] : ug[1]);
            const int nc0 = t == 0 ? c0 : (last_chunk ? 0 : c0 + F2_CH);
            const unsigned wbase = sF0 + (unsigned)((1 - t) * WQ_FB) + wofs;      // the OTHER footprint
            unsigned vb0 = E_INVALID, vb1 = E_INVALID;
            int erows = 0;
            if (EP) { vb0 = epi_base(etile, 0); vb1 = epi_base(etile, 1); erows = tile_rows_of(etile); if (OUT_HL) { e_lim = erows - wrow; e_inv = E_INVALID; asm volatile("" : "+v"(e_inv)); } }
#pragma unroll
            for (int v = 0; v < NT; ++v) {
                const int cs = (v + t) & 1, ns = cs ^ 1;
                const bool last = v + 1 == NT;
                const int ky1 = (v + 1) / KW, kx1 = (v + 1) % KW;        // the tap whose fragments this step reads
                if (last && !X_NOBAR) {
                    // every fragment read of this block has been issued; wait for them and for this thread's stores into the
                    // other footprint, then meet: behind this barrier nobody reads footprint t or the weights any more, and
                    // footprint 1 - t is complete
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // 24 slots of ONE MFMA: s = term * 8 + rb * 2 + cb, terms a.l x b.h, a.h x b.h, a.h x b.l
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    const int term = s >> 3, rb = (s >> 1) & 3, cb = s & 1;
                    floatx16& e = rb == 0 ? (cb ? d01 : d00) : rb == 1 ? (cb ? d11 : d10) : rb == 2 ? (cb ? d21 : d20) : (cb ? d31 : d30);
                    const bf16x8& av = term == 0 ? a[cs][rb].l : a[cs][rb].h;
                    const bf16x8& bv = term == 2 ? (last ? (cb ? blast.l1 : blast.l0) : (cb ? bl.l1 : bl.l0)) : (cb ? bh[cs].h1 : bh[cs].h0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ZC && v == 0 && term == 0) {
                        floatx16 z;
#pragma unroll
                        for (int i = 0; i < 16; ++i) z[i] = 0.f;
                        e = mfma(av, bv, z);
                    } else {
                        e = mfma(av, bv, e);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- slots 0..11: the fragment reads, one per slot.  Slots 0, 1: the lo weights of THIS tap (used from slot 16 on; the
                    // last tap's were read into their own set in the step before, in front of the barrier).  Slots 2..9: the A fragments
                    // of the next tap (of the next block's first tap in the last step), a.l x 4 then a.h x 4; slots 10, 11: its hi weights
                    if (!X_NOREAD && s < 12) {
                        const unsigned bcur = bread + (unsigned)(v * F2_BST);
                        const unsigned bnxt = bread + (unsigned)((last ? 0 : v + 1) * F2_BST);
                        const bool bok = !last || t == 0;            // (t == 1: the next chunk's weights are still in flight)
                        if (!last) {
                            // running addresses of the next tap's filter row (one v_add per row block when the row changes)
                            if (s == 2 && kx1 == 0 && ky1 >= 2) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    if (ky1 & 1) { asm volatile("" : "+v"(ro[r])); ro[r] += wstep2; }
                                    else { asm volatile("" : "+v"(re[r])); re[r] += wstep2; }
                                }
                            }
                        } else if (s == 2) {         // lane bases of the next block's tile
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const unsigned nbr = t == 0 ? lb[1][r] : (last_chunk ? lbn[r] : lb[0][r]);
                                re[r] = nbr; ro[r] = (nbr ^ 32u) + (unsigned)(p.W * WQ_ROW);
                            }
                        }
                        const int ky = last ? 0 : ky1, kx = last ? 0 : kx1;
#define ISS_WQ_RA(r) ((ky & 1) ? ro[r] : re[r])
                        if (s == 0 && !last) bl.l0 = *(LdsR16)(bcur + 2048);
                        else if (s == 1 && !last) bl.l1 = *(LdsR16)(bcur + 3072);
                        else if (s >= 2 && s < 6) read_a_l(a[ns][s - 2], ISS_WQ_RA(s - 2), kx);
                        else if (s >= 6 && s < 10) read_a_h(a[ns][s - 6], ISS_WQ_RA(s - 6), kx);
                        else if (s == 10 && bok) bh[ns].h0 = *(LdsR16)(bnxt);
                        else if (s == 11 && bok) bh[ns].h1 = *(LdsR16)(bnxt + 1024);
#undef ISS_WQ_RA
                    }
                    // the last tap's lo weights, in the step before it (slots 12, 13)
                    if (!X_NOREAD && v + 2 == NT && (s == 12 || s == 13)) {
                        const unsigned bl14 = bread + (unsigned)((NT - 1) * F2_BST);
                        if (s == 12) blast.l0 = *(LdsR16)(bl14 + 2048); else blast.l1 = *(LdsR16)(bl14 + 3072);
                    }
                    if (X_NOREAD && s == 5) {
                        asm volatile("" : "+v"(a[ns][0].h), "+v"(a[ns][1].h), "+v"(a[ns][2].h), "+v"(a[ns][3].h), "+v"(bh[ns].h0), "+v"(bl.l0), "+v"(blast.l0));
                    }
                    // ---- the last step of a t == 1 block: the next chunk's weights, one LDS-DMA per slot (slots 0..14)
                    if (last && t == 1 && !X_NOBAR && s < NT) load_weight_tap(nc0, s);
                    // ---- footprint pipeline.  Slice q is converted in step q + 3 (q <= 10, six pieces in slots 12..17; slices 11 and
                    // 12 beside slices 8 and 9 in steps 11 and 12, slots 18..23) and fetched three steps earlier (three pieces in
                    // slots 18..20; slices 11 and 12 in slots 21..23 of steps 8 and 9): at most four slices wait in registers
                    if (!X_NOPIPE) {
                        if (v == 0 && s == 12) wx = settle(wpend);
                        if (v == 0 && s == 13) fetch_consts(nc0);
                        if (v == 13 && s == 18) wpend = windows_of(un2.wb);
                        if (v <= 10 && s >= 18 && s <= 20) {
                            if (s == 18) fetch_1(un, v);
                            if (s == 19) fetch_2(v);
                            if (s == 20) fetch_3(nc0, v);
                        }
                        if (v >= 8 && v <= 9 && s >= 21) {
                            if (s == 21) fetch_1(un, v + 3);
                            if (s == 22) fetch_2(v + 3);
                            if (s == 23) fetch_3(nc0, v + 3);
                        }
                        if (v == 2 && s == 12) conv_consts_1();
                        if (v == 2 && s == 13) conv_consts_2();
                        if (v == 2 && s == 14) conv_consts_3();
                        if (v >= 3 && v <= 13 && s >= 12 && s <= 17) {
                            const int q = v - 3;
                            if (s == 12) convert_1(cva, q);
                            if (s == 13) convert_2(cva, q);
                            if (s == 14) convert_3(cva);
                            if (s == 15) convert_4(cva);
                            if (s == 16) convert_5(cva);
                            if (s == 17) convert_6(cva, q, wbase);
                        }
                        if (v >= 11 && v <= 12 && s >= 18) {
                            const int q = v;
                            if (s == 18) convert_1(cvb, q);
                            if (s == 19) convert_2(cvb, q);
                            if (s == 20) convert_3(cvb);
                            if (s == 21) convert_4(cvb);
                            if (s == 22) convert_5(cvb);
                            if (s == 23) convert_6(cvb, q, wbase);
                        }
                    }
                    // ---- epilogue of the other accumulator set: 32 (accumulator, window) units of three pieces; slots 0..8 of the
                    // steps 1..11 carry one piece each beside their fragment read (three units per step)
                    if (!OUT_HL && EP && v >= 1 && v <= 11 && s < 9) {
                        const int unit = (v - 1) * 3 + s / 3;                  // 0..32
                        if (unit < 32) {
                            const int erb = unit >> 3, ecb = (unit >> 2) & 1, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb ? o01 : o00) : erb == 1 ? (ecb ? o11 : o10) : erb == 2 ? (ecb ? o21 : o20) : (ecb ? o31 : o30);
                            if (s % 3 == 0) epi_a(oa, eg);
                            else if (s % 3 == 1) epi_b(oa, ecb, eg);
                            else epi_c(erb, ecb, eg, vb0, vb1, erows);
                        }
                    }
                    // OUT_HL: four pieces per unit (+ the operand split), slots 0..11
                    if (OUT_HL && EP && v >= 1 && v <= 11 && s < 12) {
                        const int unit = (v - 1) * 3 + s / 4;
                        if (unit < 32) {
                            const int erb = unit >> 3, ecb = (unit >> 2) & 1, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb ? o01 : o00) : erb == 1 ? (ecb ? o11 : o10) : erb == 2 ? (ecb ? o21 : o20) : (ecb ? o31 : o30);
                            if (s % 4 == 0) epih_a(oa, eg);
                            else if (s % 4 == 1) epih_b(oa, ecb, eg);
                            else if (s % 4 == 2) epi_s();
                            else epih_c(erb, ecb, eg, vb0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t == 1 && !X_NOBAR) {                // chunk boundary: the weights must have landed before anybody reads them
                if (!X_NOCHUNKBAR) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
                BHi& f = bh[(NT + t) & 1];
                f.h0 = *(LdsR16)(bread); f.h1 = *(LdsR16)(bread + 1024);
            }
        };
#define ISS_WQ_SET0 c000, c001, c010, c011, c020, c021, c030, c031
#define ISS_WQ_SET1 c100, c101, c110, c111, c120, c121, c130, c131
        // first chunk: zero-C; tile 0's block also drains the PREVIOUS group's tile 1 (set 1), which it does not touch
        run_block(0, true, true, 0, false, ISS_WQ_SET0, ISS_WQ_SET1, prev_tile1);
        run_block(1, true, false, 0, false, ISS_WQ_SET1, ISS_WQ_SET0, 0);
        for (int ch = 1; ch + 1 < nchunk; ++ch) {
            run_block(0, false, false, ch * F2_CH, false, ISS_WQ_SET0, ISS_WQ_SET1, 0);
            run_block(1, false, false, ch * F2_CH, false, ISS_WQ_SET1, ISS_WQ_SET0, 0);
        }
        {   // last chunk: geometry of what follows this group (uniform part + tile 0's lane bases); tile 1's block drains tile 0
            if (!last_group) {
                const GeoArgs ga = geo_args();
                ug[G] = geo_uniform(ga, (grp + gstep) * G);
                ug[G + 1] = geo_uniform(ga, (grp + gstep) * G + 1);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) lbn[rb] = geo_lane(ga, (grp + gstep) * G, rb, ug[G], 0);
            } else {
                ug[G] = ug[0]; ug[G + 1] = ug[1];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) lbn[rb] = lb[0][rb];
            }
            run_block(0, false, false, (nchunk - 1) * F2_CH, true, ISS_WQ_SET0, ISS_WQ_SET1, 0);
            run_block(1, false, true, (nchunk - 1) * F2_CH, true, ISS_WQ_SET1, ISS_WQ_SET0, grp * G);
        }
        prev_tile1 = grp * G + 1;
        if (!last_group) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) lb[0][rb] = lbn[rb];
            ug[0] = ug[G]; ug[1] = ug[G + 1];
            const GeoArgs ga = geo_args();
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) lb[1][rb] = geo_lane(ga, (grp + gstep) * G + 1, rb, ug[1], 1);
        }
    }
    // ---- the last group's tile 1: the only serial epilogue of the workgroup
    {
        const unsigned vb0 = epi_base(prev_tile1, 0), vb1 = epi_base(prev_tile1, 1);
        const int erows = tile_rows_of(prev_tile1);
#pragma unroll
        for (int unit = 0; unit < 32; ++unit) {
            const int erb = unit >> 3, ecb = (unit >> 2) & 1, eg = unit & 3;
            const floatx16& oa = erb == 0 ? (ecb ? c101 : c100) : erb == 1 ? (ecb ? c111 : c110) : erb == 2 ? (ecb ? c121 : c120) : (ecb ? c131 : c130);
            if (OUT_HL) { e_lim = erows - wrow; e_inv = E_INVALID; asm volatile("" : "+v"(e_inv)); epih_a(oa, eg); epih_b(oa, ecb, eg); epi_s(); epih_c(erb, ecb, eg, vb0); }
            else { epi_a(oa, eg); epi_b(oa, ecb, eg); epi_c(erb, ecb, eg, vb0, vb1, erows); }
        }
    }
#undef ISS_WQ_SET0
#undef ISS_WQ_SET1
}

// host: does the weight-stationary quad kernel take this launch?  (fused first layer, unpadded, relu + 2 x 2 max-pool, 64 output
// channels per workgroup, 5x3; tiles of ConvArgs::tmr rows within 800 pixels)
inline bool iss_wq_compiled(int kh, int kw) { return kh == 5 && kw == 3; }
void iss_wq_launch_5x3(const ConvArgs& a, dim3 grid, hipStream_t st);       // a.out_hl: the CHL-output instantiation

}  // namespace issk
