// conv_x3_ws_kernel: weight-stationary LDS-footprint bf16x3 implicit GEMM (8 <= kh*kw <= 16 taps, Cin % 16 == 0,
// one 64-column N tile per workgroup).
//
// The footprint kernel of conv_fp.h streams an 8 KB weight tile per 12 MFMAs and wave through LDS (LDS-DMA,
// a ring of stages, a counted vmcnt wait and a workgroup barrier per tile): for the 5x3 64->64 convolution that is 63 % of
// a segmenter step the weights of one tile (245 KB) are 3.5 x its input footprint, and they are the SAME for every tile.
// rocprofv3 counters (profiles/r01_pmc.md) showed those kernels at 47 % matrix-pipe occupancy with 30 % of the wave
// cycles parked at barriers / waits and 39 % issue-stalled.  Here the loop nest is turned around:
//
//   for each group of G = 2 tiles (512 rows each: 8 waves x 64 rows x 64 channels; 2 x 4 accumulators = 128 VGPRs)
//     for each 16-channel chunk c
//       load the weights of ALL taps of chunk c into LDS once (NT x 4 KB, LDS-DMA)            <- once per 2 tiles
//       for each tile t of the group                                                          <- one "block"
//         NT steps of 12 MFMAs in three pinned groups (a.lo x b.hi, a.hi x b.hi, a.hi x b.lo on 2 row x 2 column blocks):
//         A fragments from the tile's LDS footprint (double-buffered in registers), B fragments from the resident weights
//         (single-buffered: the hi plane of step v + 1 is read behind the second group of step v, the lo plane behind the
//         third) -- 8 ds_read_b128 per 12 MFMAs; behind them the footprint of the NEXT block is fetched, split into bf16
//         hi / lo and kept in the registers of the f32 values it replaces; at the end of the block: barrier, the footprint
//         is written (14 ds_write_b64 per thread), barrier (180 MFMAs per wave and block for a 5x3 filter).
//
// No weight ring, no vmcnt bookkeeping and no barrier inside a block: a wave's stream is ds_read_b128 + MFMA with a
// conversion slice per step in the second half of the block.  512 threads = 8 waves (two per SIMD, decoupled between the
// block boundaries) share ONE footprint of 896 pixels (72 KB; two of them do not fit beside the weights) and the weight
// block (60 KB for 5x3).  The 32-row-per-wave predecessor (four 256-row tiles per group, two 512-pixel footprints, one
// barrier per block) needed 12 fragment reads per 12 MFMAs and ran 8-10 % slower.
//
// LDS layout.  Footprint rows are 80 bytes per pixel -- 16 channels hi (32 B) | 16 channels lo (32 B) | 16 B pad -- and
// LINEAR: the A-fragment address of a tap is (lane base + tap offset), one v_add per step, and 16 consecutive pixels still
// hit 16 different 16-byte bank groups (20 p mod 64 is a permutation of the multiples of 4).  Zero-padded taps read an
// all-zero pixel kept behind the footprint (one v_cndmask on the address instead of zeroing eight fragment registers).
// The 16-byte slots of a weight tile are permuted on the SOURCE side of the LDS-DMA (slot = 2 n + (h ^ ((n >> 3) & 1)) for
// row n, k half h) so that the ds_read_b128 of the 16 lanes of a group are conflict-free.  Geometry and epilogue
// parameters are read through an opaque copy of the kernel-argument pointer where they are needed instead of living in
// SGPRs through the main loop (conv_x3_fp_kernel spills 136 SGPRs and 11 VGPRs of its 45-word argument block).
#pragma once
#include "conv_fp.h"

namespace issk {

constexpr int F2_ROW = 80;                        // bytes per footprint pixel: 16 ch hi (32 B) | 16 ch lo (32 B) | 16 B pad
constexpr int F2_BST = 4096;                      // bytes of one weight stage: hi plane (64 rows x 16 k bf16 = 2 KB) | lo plane
constexpr int F2_CH = 16;                         // channels per chunk (one k16 MFMA step)

typedef const bf16x8 __attribute__((address_space(3)))* LdsR16;
typedef bf16x4 __attribute__((address_space(3)))* LdsW8;
typedef unsigned __attribute__((address_space(3)))* LdsW4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 __attribute__((address_space(3)))* LdsF4;

// geometry parameters (row decomposition of a tile): loaded from the kernel-argument segment once per tile
struct GeoArgs {
    int H, W, Hq, Wq, ph, pw, pp, sh, sw, pt_, pl_;
    unsigned dv_mul[4];
    int dv_sh[4];
};
typedef const ConvArgs __attribute__((address_space(4)))* KArg;

// epilogue parameters, loaded from the kernel-argument segment when a tile is complete
struct EpiArgs {
    const float* bias; const float* ps; const float* pt; const float* res; float* out;
    long long M; int Cout, act, pp, poolkind;
    unsigned out_np; int out_f16;                  // CHL output of the pooled epilogue (conv_common.h chl_store), 0 = f32 NHWC
};


constexpr int WS_TM = 512;                         // GEMM rows per tile: 8 waves x 64 (two 32-row blocks per wave)
constexpr int WS_G = 2;                            // tiles per group (NH = 1)
constexpr int WS_PIX = 896;                        // footprint capacity in pixels, NH = 1 (host-validated per launch)
constexpr int WS_PIX2 = 1024;                      // footprint capacity, NH = 2 (9-tap layers: 18 x 4 KB of weights beside it)
constexpr int WS_MAXNT = 16;                       // taps: NT x 4 KB of weights + the footprint must fit 160 KB of LDS
// RING (filters with more than WS_MAXNT taps, e.g. 7x7 = 49; NH = 1 only): the weights of a 16-channel chunk do not fit LDS at
// once (49 x 4 KB), so the taps are taken in groups of TG and the resident weights become a ring of two halves of TG tiles:
// group k lives in half k % 2, and while a wave runs the steps of group k the tiles of group k + 1 land in the other half
// (LDS-DMA, issued right behind the barrier that retires group k - 1 -- one barrier per TG steps = 12 TG MFMAs per wave,
// where conv_x3_fp_kernel's per-tap ring pays one per 12).  TG is the largest value <= 9 that gives an EVEN number of groups,
// so that every block starts with group 0 in half 0 and all ring slots stay compile-time constants.  One 512-row tile per
// group and the 1024-pixel footprint of NH = 2 (a 512-row tile under a 7-row filter spans 920-1020 pixels).
constexpr bool ws_ring(int nt) { return nt > WS_MAXNT; }
constexpr int ws_tg(int nt) {
    for (int tg = 9; tg >= 5; --tg)
        if (((nt + tg - 1) / tg) % 2 == 0) return tg;
    return 0;
}
constexpr int ws_pix(int nh, int nt = 0) { return (nh == 2 || ws_ring(nt)) ? WS_PIX2 : WS_PIX; }
constexpr int ws_resident(int nt, int nh) { return ws_ring(nt) ? 2 * ws_tg(nt) : nt * nh; }      // 4 KB weight tiles kept in LDS
constexpr int WS_STAB = 8192;                     // FS: bytes of the first layer's per-column weight-sum table S[W][Cin] (f32)
constexpr int ws_lds_bytes(int nt, int nh, bool fs = false) {
    return (ws_pix(nh, nt) + 1) * F2_ROW + ws_resident(nt, nh) * F2_BST + (fs ? WS_STAB : 0);
}

// NH = 2 (layers with >= 128 output channels, not first-layer-fused): the workgroup computes TWO 64-column halves of ONE
// 512-row tile per group instead of one half of two tiles -- the same eight accumulators, indexed (half, row block, column
// block) instead of (tile, row block, column block).  A block then runs NT x 2 'virtual steps' (tap, half) on one footprint
// with 2 NT resident weight tiles: the footprint is fetched / converted / written once per 128 output channels, the A
// fragments of a tap are read once for both halves (6 instead of 8 ds_read_b128 per 12 MFMAs), and a block is twice as long
// against the same serial block boundary.  This is what moved the 3x3 layers of the segmenter nets (64 -> 128, 128 -> 128)
// off conv_x3_fp_kernel (8 KB weight ring, a barrier per 12 MFMAs, 12 reads per 12 MFMAs: 42 / 55 % matrix-pipe occupancy).
// EPI = 1: the pooled relu epilogue only (epilogue_pool_relu; TR: bias + optional relu only, epilogue_tr<.., SIMPLE>); host-checked.
// EPI = 0: the generic ones.
// FS (FUSED only): the first layer in front of this convolution is zero-PADDED ('same').  Its padding is applied to the
// z-normalised window, i.e. a padded position is worth the window's mean in raw log-mel units, so two things depend on where a
// first-layer output sits in its window: (i) its COLUMN x decides which filter columns saw data -- the weight sum of the
// per-window affine map becomes a table S[x][c] (ConvArgs::f_wsum, [W][Cin] f32, <= 8 KB, kept in LDS; two FMAs per value
// instead of one); (ii) the first f_padt / last f_padb ROWS of a window saw fewer filter rows: those few rows are not shared between
// windows -- first_layer_edge_kernel writes them per window, already corrected so that the SAME affine map applies
// (E = partial sum + mean_b * (S - S_partial)), behind the shared rows in `in`, and only their address differs in the fetch.
// F32 (ISS_PREC_F32, the exact-f32 mode): the same kernel on v_mfma_f32_32x32x2_f32.  LDS holds f32 instead of bf16 hi | lo -- the
// SAME bytes per element, so footprint, weight tiles, addresses and the whole pipeline are unchanged: a pixel's 64 data bytes are
// 16 floats (k quads 0..3), a weight tile is 64 rows x 16 floats with quads {0, 1} where the hi plane was and {2, 3} where the lo
// plane was.  A lane's two 16-byte fragment reads give it quads lh and 2 + lh; MFMA j of a quad multiplies element j of the A quad
// with element j of the B quad, i.e. lanes 0-31 contribute k = 4 q + j and lanes 32-63 k = 4 (q + 1) + j -- a permutation of the k
// order that both operands share.  8 MFMAs (16 passes each) per 32 x 32 block and k-step of 16 instead of three bf16 ones; no
// operand split in the conversion.  Bit-wise an fmaf chain per output, like conv_igemm_kernel, in a different k order.
// NCB = 1 (layers with <= 32 output channels; row-major epilogue, NH = 1): one 32-column block per workgroup instead of two -- a
// 32-channel layer on the 64-column form spends half of its MFMAs on columns that do not exist.  6 MFMAs per tap instead of 12, the
// weight rows 32..63 of a tile are neither fetched nor read.
template <int KH, int KW, bool PADDED, bool TR, bool FUSED, int NH = 1, int EPI = 0, bool FS = false, bool F32 = false, int NCB = 2>
__global__ __launch_bounds__(512, 2) void conv_x3_ws_kernel(const ConvArgs p) {
    static_assert(NCB == 2 || (NCB == 1 && !TR && NH == 1), "");
    constexpr int NT = KH * KW;
    constexpr int NV = NT * NH;                      // virtual steps per block
    constexpr bool RING = ws_ring(NT);               // more taps than resident weight tiles: ring of two tap groups (see ws_tg)
    constexpr int TG = RING ? ws_tg(NT) : NV;        // steps per tap group
    constexpr int NG = RING ? (NV + TG - 1) / TG : 1;
    constexpr int NRES = ws_resident(NT, NH);        // resident 4 KB weight tiles
#ifdef ISS_WS_FS_G2                                  // comparison build only (profiles/r06_scripts/r06_fs_g_ab.sh): the FS form with two tiles per group, as in round 5
    constexpr int G = (NH == 2 || RING) ? 1 : WS_G;
#else
    constexpr int G = (NH == 2 || RING || FS) ? 1 : WS_G;
#endif  // tiles per group (FS: one -- its 64 fewer accumulator registers are what the S-table arithmetic needs)
    constexpr int PIX = ws_pix(NH, NT);
    constexpr int WS_ZERO = PIX * F2_ROW;            // byte offset of the all-zero pixel behind the footprint
    constexpr int WS_NFV = PIX / 128;                // 128-pixel slices per footprint: 512 threads x 4 channels each
    static_assert(NT >= 8 && (NH == 1 || (NH == 2 && !FUSED && !RING)) && (!FS || FUSED), "");
    static_assert(!RING || (TG >= 5 && NG % 2 == 0 && NG >= 2), "");
    static_assert(ws_lds_bytes(NT, NH, FS) <= 160 * 1024, "");
    // [resident weight tiles][footprint][FS: S table]: the weights first, so that (lane base + slot * 4096 + plane / half offset)
    // stays an immediate offset of ds_read for the first 16 tiles and the steps share ONE address register
    __shared__ __attribute__((aligned(1024))) unsigned char smem[ws_lds_bytes(NT, NH, FS)];
    const unsigned sB_base = (unsigned)(size_t)smem;
    const unsigned sF_base = sB_base + NRES * F2_BST;
    const unsigned sS_base = sF_base + (unsigned)((PIX + 1) * F2_ROW);
    auto slot_of = [](int v) { return RING ? ((v / TG) & 1) * TG + v % TG : v; };      // LDS slot of weight tile v

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..7
    const int n0 = blockIdx.y * (NCB == 1 ? 32 : BN) * NH;
    const int li = lane & 31, lh = lane >> 5;
    const int M = (int)p.M;
    int totpix;                                      // samples * H * W
    { const int spp = p.Hq * p.Wq * p.pp; totpix = (int)(p.img_stride / p.Cin) * (M / spp); }
    // RING: the host picks the rows per tile (ConvArgs::tmr <= 512, a multiple of 4) so that a tile's footprint fits PIX pixels under a
    // tall filter; the rows of a tile beyond tmr are computed on clamped addresses and not stored
    const int TMR = RING ? p.tmr : WS_TM;
    const int ntiles = (M + TMR - 1) / TMR;
    const int ngroups = (ntiles + G - 1) / G;
    int grp = (int)blockIdx.x;
    if (grp >= ngroups) return;

    // ---- geometry (row decomposition parameters through the kernel-argument pointer)
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
    struct TGeo { int p_lo, need, fy, fx, wb; };     // uniform per tile: first pixel, pixels needed, FUSED: (y, x) of it and its window
    auto clamp_tile = [&](int t) { return t < ntiles ? t : ntiles - 1; };
    auto geo_uniform = [&](const GeoArgs& ga, int tile) {
        TGeo u;
        const int m0 = clamp_tile(tile) * TMR;
        int b, oy, ox;
        map_row32(ga, m0, b, oy, ox);
        u.p_lo = (b * ga.H + (oy * ga.sh - ga.pt_)) * ga.W + (ox * ga.sw - ga.pl_);
        u.fy = oy * ga.sh - ga.pt_; u.fx = ox * ga.sw - ga.pl_; u.wb = b;     // (negative for the padding rows / pixels)
        const int ml = m0 + TMR - 1 < M - 1 ? m0 + TMR - 1 : M - 1;
        int b2, oy2, ox2;
        map_row32(ga, ml, b2, oy2, ox2);
        u.need = (b2 * ga.H + (oy2 * ga.sh - ga.pt_ + KH - 1)) * ga.W + (ox2 * ga.sw - ga.pl_ + KW - 1) - u.p_lo + 1;
        return u;
    };
    using VMask = typename std::conditional<(NT > 32), unsigned long long, unsigned>::type;      // one validity bit per tap
    auto geo_lane = [&](const GeoArgs& ga, int tile, int rb, const TGeo& u, int& lanepix, VMask& vmask) {   // lanepix: see read_a
        const int m0 = clamp_tile(tile) * TMR;
        const int m = m0 + (wv * 2 + rb) * 32 + li;
        int b, oy, ox;
        map_row32(ga, (m < M && (!RING || m < m0 + TMR)) ? m : m0, b, oy, ox);
        const int iy0 = oy * ga.sh - ga.pt_, ix0 = ox * ga.sw - ga.pl_;
        const int lp = (b * ga.H + iy0) * ga.W + ix0 - u.p_lo;
        const int hi = PIX - 1 - ((KH - 1) * ga.W + (KW - 1));       // keeps every tap of a row >= M inside the buffer
        lanepix = (int)sF_base + (lp < 0 ? 0 : (lp > hi ? hi : lp)) * F2_ROW + lh * 16;      // LDS byte address of the lane's first tap
        vmask = ~(VMask)0;
        if (PADDED) {
            VMask vm = 0;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
                    vm |= ((unsigned)(iy0 + ky) < (unsigned)ga.H && (unsigned)(ix0 + kx) < (unsigned)ga.W) ? (VMask)1 << (ky * KW + kx) : (VMask)0;
            vmask = vm;
        }
    };
    // FUSED: per-window scalars of the (at most two) windows a footprint touches.  Loaded one block before they are used
    // (`wpend`), then moved to SGPRs (`settle`): they are wave-uniform, and VGPRs are what this kernel is short of.
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

    // ---- weights of one 16-channel chunk: NT tiles of 4 KB = 4 NT pieces of 1 KB; wave w moves pieces w, w + 8, ...
    // piece i: tap i >> 2, plane (i >> 1) & 1 (hi / lo), half i & 1 (rows 0-31 / 32-63).  Lane l of a piece writes 16-byte
    // slot 64 half + l and fetches the (row n, k half h) that belongs there: n = 32 half + (l >> 1),
    // h = (l & 1) ^ ((n >> 3) & 1)  (conflict-free B reads, see above).  Rows >= Cout read row 0 (never stored).
    // NH = 2: weight tile v = tap * 2 + column half ch; the rows of column half ch start at n0 + 64 ch.  Piece i = wv + 8 k
    // has (half, plane, ch) = (wv & 1, (wv >> 1) & 1, (wv >> 2) & 1) for EVERY k: a wave always moves the same kind of piece,
    // only the tap changes, so one source offset per lane serves all of its pieces.
    const int w_half = wv & 1, w_plane = (wv >> 1) & 1, w_ch = NH == 2 ? (wv >> 2) & 1 : 0;
    unsigned boff_w;
    {
        const int n = 32 * w_half + (lane >> 1), h = (lane & 1) ^ ((n >> 3) & 1);
        const int row = n0 + 64 * w_ch + n;
        if (F32) boff_w = 4u * ((unsigned)(row < p.Cout ? row : 0) * (unsigned)p.Kpad + (unsigned)(h * 4));     // bytes: quad 2 plane + h
        else boff_w = 2u * ((unsigned)(row < p.Cout ? row : 0) * (unsigned)p.Kpad + (unsigned)(h * 8));      // bytes
    }
    // tiles [v_lo, v_hi) of chunk c0
    auto load_weights = [&](int c0, int v_lo, int v_hi) {
#pragma unroll
        for (int k = 0; k < (4 * NV + 7) / 8; ++k) {
            const int i = wv + 8 * k;                // uniform
            if (i < 4 * NV && (i >> 2) >= v_lo && (i >> 2) < v_hi && !(NCB == 1 && w_half)) {
                const int v = i >> 2;                // weight tile: tap v / NH (column half v % NH == w_ch)
                const uint16_t* src = F32 ? reinterpret_cast<const uint16_t*>(p.w + ((v / NH) * p.Cin + c0 + w_plane * 8))
                                          : (w_plane ? p.wl : p.wh) + ((v / NH) * p.Cin + c0);
                if (RING) {                          // (the 49-step form ran out of SGPRs and kept this uniform address in VGPRs: back into an SGPR pair)
                    const unsigned long long a64 = (unsigned long long)src;
                    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a64);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a64 >> 32));
                    src = (const uint16_t*)(((unsigned long long)hi << 32) | lo);
                }
                glds16(src, boff_w,
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_base + slot_of(v) * F2_BST + w_plane * 2048 + w_half * 1024)));
            }
        }
    };
    unsigned bread = sB_base + (unsigned)((2 * li + (lh ^ ((li >> 3) & 1))) * 16);
    asm volatile("" : "+v"(bread));                  // opaque base: the per-tap offsets stay immediates (no per-tap address registers)

    // ---- footprint slices: thread -> pixel 128 q + (tid >> 2), channels [c0 + 4 (tid & 3), + 4)
    const int cg = tid & 3, prow = tid >> 2;
    float4 fv[WS_NFV];
    unsigned dbmask = 0;
    float4 fsw = make_float4(0.f, 0.f, 0.f, 0.f), fbw = fsw;     // FUSED: weight sums / bias of this thread's 4 first-layer channels
    Win wx = {}, wpend = {};                         // windows of the footprint being built (settled) / of the one after it (pending)
    const int magicW = (65536 + p.W - 1) / p.W;      // x / W == (x * magicW) >> 16 for x < 65536 / W (host-checked for FUSED)
    const unsigned cin4 = (unsigned)p.Cin * 4u, cg16 = (unsigned)cg * 16u;         // byte strides (32-bit offsets from a uniform base)
    int fs_x0 = 0;                                   // FS: flattened (y * W + x) position of this thread's slice-0 pixel in its window
    unsigned fs_sc = 0;                              // FS: LDS address of S[0][c0 + 4 cg] for the chunk being built
    auto fetch_block = [&](const TGeo& u, int c0) {  // all loads of one footprint chunk (FUSED: with wx = its windows' scalars)
        if (FUSED) {
            const unsigned o = (unsigned)(c0 + cg * 4);
            if (!FS) fsw = *reinterpret_cast<const float4*>(p.f_wsum + o);
            fbw = *reinterpret_cast<const float4*>(p.f_bias + o);
            if (FS) { fs_x0 = u.fy * p.W + u.fx + prow; fs_sc = sS_base + o * 4u; }
        }
        dbmask = 0;
        int f_padt = 0, f_ne = 0, f_ybot = 0, f_erow0 = 0;      // FS: through the kernel-argument pointer (not resident in SGPRs)
        if (FS) {
            KArg qa = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(qa));
            f_padt = qa->f_padt; f_ne = f_padt + qa->f_padb; f_ybot = qa->H - qa->f_padb; f_erow0 = qa->f_erow0;
        }
#pragma unroll
        for (int q = 0; q < WS_NFV; ++q) {
            const int qq = 128 * q < u.need ? q : 0; // unneeded slices re-load slice 0
            if (FUSED) {
                // pixel of the footprint inside its window, flattened (y * W + x).  With a zero-padded convolution the
                // footprint starts up to pt rows / pl pixels BEFORE the window's first pixel: those positions are only ever
                // read by masked taps (they fetch pixel 0 instead)
                int x = u.fy * p.W + u.fx + prow + 128 * qq;
                x = PADDED && x < 0 ? 0 : x;
                const int dy = (x * magicW) >> 16;
                x -= dy * p.W;
                int y = dy;
                const bool second = y >= p.H;
                y -= second ? p.H : 0;
                dbmask |= second ? 1u << q : 0u;
                int row = y + (second ? wx.wr1 : wx.wr0) - p.f_rmin;
                if (FS) {                            // the first f_padt / last f_padb rows of a window: its own edge rows behind the shared ones
                    const int yb = y - f_ybot;
                    const int e = y < f_padt ? y : (yb >= 0 ? f_padt + yb : -1);
                    int bw = u.wb + (second ? 1 : 0);
                    bw = bw < nwin ? bw : nwin - 1;
                    row = e >= 0 ? f_erow0 + bw * f_ne + e : row;
                }
                fv[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.in + c0) + ((unsigned)(row * p.W + x) * cin4 + cg16));
            } else {
                int gp = u.p_lo + prow + 128 * qq;
                gp = gp < 0 ? 0 : (gp > totpix - 1 ? totpix - 1 : gp);
                fv[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.in + c0) + ((unsigned)gp * cin4 + cg16));
            }
        }
    };
    const float f_lob = p.f_act == 1 ? 0.f : -INFINITY;
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f}, rs0 = 0.f, rs1 = 0.f;
    float fs_mr0 = 0.f, fs_mr1 = 0.f;               // FS: -mean / std of the two windows (the shift is fma(S[x][c], mr, bias) per value)
    auto conv_consts = [&]() {
        if (!FUSED) return;
        if (FS) asm volatile("" : "+v"(fbw.x), "+v"(fbw.y), "+v"(fbw.z), "+v"(fbw.w));
        else asm volatile("" : "+v"(fsw.x), "+v"(fsw.y), "+v"(fsw.z), "+v"(fsw.w), "+v"(fbw.x), "+v"(fbw.y), "+v"(fbw.z), "+v"(fbw.w));
        rs0 = wx.live0 ? 1.0f / wx.sd0 : 0.f;
        rs1 = wx.live1 ? 1.0f / wx.sd1 : 0.f;
        const float mr0 = wx.live0 ? -wx.mean0 * rs0 : 0.f, mr1 = wx.live1 ? -wx.mean1 * rs1 : 0.f;
        if (FS) { fs_mr0 = mr0; fs_mr1 = mr1; return; }
        const float sw[4] = {fsw.x, fsw.y, fsw.z, fsw.w}, bw[4] = {fbw.x, fbw.y, fbw.z, fbw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { t0[i] = fmaf(sw[i], mr0, bw[i]); t1[i] = fmaf(sw[i], mr1, bw[i]); }
    };
    // The next footprint is converted BEHIND the MFMAs of the current block but kept in registers (bf16 hi / lo of the
    // thread's 4 channels: 2 + 2 VGPRs per slice, taking over the 4 registers of the slice's f32 values); it is written into
    // the (single) LDS footprint at the block boundary, between two barriers.  Two footprints of 896 pixels would not fit
    // beside the resident weights.
    bf16x4 cvh[WS_NFV], cvl[WS_NFV];
    auto convert_slice = [&](int q) {                // q: compile-time
        float4 v = fv[q];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(dbmask));    // not before this point (see fetch_block)
        if (FUSED && FS) {
            // column of this slice's pixel -> its row of the S table (the position inside the window wraps at W; a pixel of the
            // second window has the same column arithmetic because windows are whole rows)
            int xf = fs_x0;
            asm volatile("" : "+v"(xf));             // computed HERE, per slice: hoisted to the fetch it costs a live register per slice
            xf += 128 * q;
            xf = PADDED && xf < 0 ? 0 : xf;
            const int xx = xf - ((xf * magicW) >> 16) * p.W;
            const f32x4 sv = *(LdsF4)(fs_sc + (unsigned)xx * cin4);
            const bool second = (dbmask >> q) & 1u;
            const float sc = second ? rs1 : rs0, mr = second ? fs_mr1 : fs_mr0;
            v = make_float4(fmaf(v.x, sc, fmaf(sv.x, mr, fbw.x)), fmaf(v.y, sc, fmaf(sv.y, mr, fbw.y)),
                            fmaf(v.z, sc, fmaf(sv.z, mr, fbw.z)), fmaf(v.w, sc, fmaf(sv.w, mr, fbw.w)));
            v.x = fmaxf(v.x, f_lob); v.y = fmaxf(v.y, f_lob); v.z = fmaxf(v.z, f_lob); v.w = fmaxf(v.w, f_lob);
        } else if (FUSED) {
            const bool second = (dbmask >> q) & 1u;
            const float sc = second ? rs1 : rs0;
            v = make_float4(fmaf(v.x, sc, second ? t1[0] : t0[0]), fmaf(v.y, sc, second ? t1[1] : t0[1]),
                            fmaf(v.z, sc, second ? t1[2] : t0[2]), fmaf(v.w, sc, second ? t1[3] : t0[3]));
            v.x = fmaxf(v.x, f_lob); v.y = fmaxf(v.y, f_lob); v.z = fmaxf(v.z, f_lob); v.w = fmaxf(v.w, f_lob);
        }
        if (F32) {                                   // the 4 floats as they are: (v.x, v.y) and (v.z, v.w) in the two 8-byte registers
            struct F2 { float a, b; };
            cvh[q] = __builtin_bit_cast(bf16x4, F2{v.x, v.y});
            cvl[q] = __builtin_bit_cast(bf16x4, F2{v.z, v.w});
        } else split4(v, cvh[q], cvl[q]);
    };
    // bf16: 4 channels hi at +8 cg, lo 32 bytes behind; F32: the 4 floats are one 16-byte quad at +16 cg
    constexpr int FW_SECOND = F32 ? 1 : 4;           // 8-byte units between the two stores of a slice
    const unsigned fwrite = sF_base + (unsigned)(prow * F2_ROW + cg * (F32 ? 16 : 8));
    auto write_footprint = [&]() {
        unsigned b = fwrite;
        asm volatile("" : "+v"(b));                  // opaque base: one address register, the slices are immediates (< 64 KB;
        const LdsW8 pb = (LdsW8)(b);                 // a second register from slice 6 on: the 8-slice footprint of NH = 2)
        constexpr int QB = 6;
        static_assert((QB - 1) * 128 * F2_ROW + 32 < 65536 && (WS_NFV - 1 - QB) * 128 * F2_ROW + 32 < 65536, "");
        unsigned b2 = fwrite + (unsigned)(QB * 128 * F2_ROW);
        asm volatile("" : "+v"(b2));
        const LdsW8 pb2 = (LdsW8)(b2);
#pragma unroll
        for (int q = 0; q < WS_NFV; ++q) {
            if (q < QB) {
                pb[q * (128 * F2_ROW / 8)] = cvh[q];
                pb[q * (128 * F2_ROW / 8) + FW_SECOND] = cvl[q];
            } else {
                pb2[(q - QB) * (128 * F2_ROW / 8)] = cvh[q];
                pb2[(q - QB) * (128 * F2_ROW / 8) + FW_SECOND] = cvl[q];
            }
        }
    };

    // ---- fragments
    struct AFr { bf16x8 h, l; };
    struct BH { bf16x8 b0, b1; };                    // hi (or lo) fragments of the two 32-column blocks
    // lpb: lane base of a tile = sF_base + lanepix * 80 + lh * 16 (per tile, kept in one register); the tap offset
    // (ky * W + kx) * 80 is wave-uniform and added per step.  The opaque copy keeps hipcc from materialising all KH * KW
    // addresses of a tile at once (15 address registers per tile drove the first build of this kernel into scratch).
    const unsigned zbase = sF_base + (unsigned)WS_ZERO + (unsigned)(lh * 16);
    const unsigned row_step = (unsigned)((p.W - (KW - 1)) * F2_ROW);    // from the last tap of a filter row to the first of the next
    // `cur` walks the taps: + 80 bytes within a filter row, + row_step at the end of one (a loop-carried value, so hipcc
    // cannot materialise all KH * KW addresses of a tile at once)
    auto read_a = [&](AFr& f, unsigned cur, VMask vmask, int tap) {                   // tap: compile-time
        unsigned a = cur;
        if (PADDED) a = ((vmask >> tap) & (VMask)1) ? a : zbase;
        f.h = *(LdsR16)(a);
        f.l = *(LdsR16)(a + 32);
    };
    auto next_tap = [&](unsigned cur, int tap) {     // address of tap + 1
        asm volatile("" : "+v"(cur));
        return cur + ((tap + 1) % KW == 0 ? row_step : (unsigned)F2_ROW);
    };
    auto read_bh = [&](BH& f, int tap) {             // hi plane of the tap's weight tile
        const unsigned a = bread + (unsigned)(slot_of(tap) * F2_BST);
        f.b0 = *(LdsR16)(a);
        if (NCB == 2) f.b1 = *(LdsR16)(a + 1024); else f.b1 = f.b0;
    };
    auto read_bl = [&](BH& f, int tap) {             // lo plane
        const unsigned a = bread + (unsigned)(slot_of(tap) * F2_BST);
        f.b0 = *(LdsR16)(a + 2048);
        if (NCB == 2) f.b1 = *(LdsR16)(a + 3072); else f.b1 = f.b0;
    };
    // one MFMA pair: A vector x the two column blocks of a weight plane, into the two accumulators of a row block
    auto mfma2 = [&](const bf16x8& av, const BH& b, floatx16& c0, floatx16& c1) {
        if constexpr (F32) {                             // the 16 bytes are one k quad of f32: four K = 2 MFMAs per column block
            const f32x4 a4 = __builtin_bit_cast(f32x4, av), b0 = __builtin_bit_cast(f32x4, b.b0), b1 = __builtin_bit_cast(f32x4, b.b1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (TR) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j], a4[j], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j], a4[j], c1, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b0[j], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], b1[j], c1, 0, 0, 0);
                }
            }
        } else
        if (TR) {                                        // C^T: rows = channels, columns = pixels (epilogue_tr)
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b0, av, c0, 0, 0, 0);
            if (NCB == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b1, av, c1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b.b0, c0, 0, 0, 0);
            if (NCB == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b.b1, c1, 0, 0, 0);
        }
    };

    // accumulators of the group's four tiles: named variables, not an array (an array that is passed by reference into the
    // pooled epilogue ended up in scratch memory)
    static_assert(WS_G == 2, "");
    floatx16 acc000, acc001, acc010, acc011, acc100, acc101, acc110, acc111;      // acc<tile><row block><column block>
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc000[i] = 0.f; acc001[i] = 0.f; acc010[i] = 0.f; acc011[i] = 0.f; acc100[i] = 0.f; acc101[i] = 0.f; acc110[i] = 0.f; acc111[i] = 0.f; }

    // conversion schedule inside a block of NV steps: loads at step 0, constants at step CS - 1, slice q at step CS + q * CSTRIDE
#ifndef ISS_CS15
#define ISS_CS15 7
#endif
#ifndef ISS_CS18
#define ISS_CS18 7
#endif
#ifndef ISS_EARLY
#define ISS_EARLY 1
#endif
    constexpr int CS = NV >= 16 ? ISS_CS18 : (NV >= 12 ? ISS_CS15 : 2);
    constexpr int CSTRIDE = (NV - CS) / WS_NFV >= 1 ? (NV - CS) / WS_NFV : 1;
    static_assert(CS + (WS_NFV - 1) * CSTRIDE <= NV - 1, "");
    // Early weight refresh: the weight tiles of steps < VB are dead once EVERY wave has reached step VB, and the last
    // conversion slice (whose compiler-placed vmcnt wait would otherwise also wait for the DMAs) is behind us: one extra
    // barrier there, and the next chunk's tiles [0, VB) are in flight for the rest of the block instead of being waited for
    // at the boundary.  Only when that buys at least two steps.
    constexpr int VB = CS + (WS_NFV - 1) * CSTRIDE + 1;
    constexpr bool EARLY_W = ISS_EARLY && !RING && VB + 2 <= NV;

    // ---- prologue: zero pixels; geometry of the first group; its first footprint converted serially
    if (tid < F2_ROW / 4) *(LdsW4)(sF_base + (unsigned)(WS_ZERO + tid * 4)) = 0u;
    if (FS) {                                        // S[x][c] (host-checked: W * Cin * 4 <= WS_STAB)
        for (int i = tid; i < p.W * p.Cin / 4; i += 512) {
            const float4 sw4 = reinterpret_cast<const float4*>(p.f_wsum)[i];
            f32x4 sv; sv.x = sw4.x; sv.y = sw4.y; sv.z = sw4.z; sv.w = sw4.w;
            *(LdsF4)(sS_base + (unsigned)i * 16u) = sv;
        }
        __syncthreads();                             // the serial conversion of the first footprint below reads it
    }
    TGeo ug[G + 2];                                  // the group's tiles + the next group's first two tiles
    int lanepix[G][2];
    VMask vmask[G][2];
    auto group_geometry = [&](int g0) {
        const GeoArgs ga = geo_args();
#pragma unroll
        for (int t = 0; t < G; ++t) {
            ug[t] = geo_uniform(ga, g0 * G + t);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) geo_lane(ga, g0 * G + t, rb, ug[t], lanepix[t][rb], vmask[t][rb]);
        }
    };
    group_geometry(grp);
    ug[G] = ug[0]; ug[G + 1] = ug[G == 1 ? 0 : 1];
    if (FUSED) { wx = settle(windows_of(ug[0].wb)); wpend = windows_of(ug[1].wb); }
    fetch_block(ug[0], 0);
    conv_consts();
#pragma unroll
    for (int q = 0; q < WS_NFV; ++q) convert_slice(q);
    write_footprint();

    const int nchunk = p.Cin / F2_CH;
    const int gstep = (int)gridDim.x;
    // ISS_DBG bit 2 (experiment): static priority for the second-dispatched half of the workgroup (waves 4..7 share their
    // SIMDs with waves 0..3 and lose the issue arbitration by age: MI355X_MICROARCH.md, "two waves per SIMD", item 4)
#ifdef ISS_EXPERIMENTS
    if ((p.dbg & 4) && wv >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    bool first_chunk = true;
    for (; grp < ngroups; grp += gstep) {
        const bool last_group = grp + gstep >= ngroups;
        if (G == 1 && FUSED) {                       // the windows of the next group's tile are requested two blocks ahead (un2)
            if (!last_group) ug[G] = geo_uniform(geo_args(), (grp + gstep) * G);
            else ug[G] = ug[0];
            ug[G + 1] = ug[G];
        }
        for (int ch = 0; ch < nchunk; ++ch) {
            const int c0 = ch * F2_CH;
            const bool last_chunk = ch + 1 == nchunk;
            // ---- weights of this chunk: their LDS-DMAs were issued at the boundary that ended the previous chunk's last
            // block (behind its first barrier, in flight while the footprint is written); only the very first chunk of the
            // workgroup loads them here.  The geometry of the next group's first tiles is computed here.
            const bool fc = first_chunk;
            if (fc) { load_weights(c0, 0, RING ? 2 * TG : NV); first_chunk = false; }
            if (last_chunk) {
                if (!last_group) {
                    const GeoArgs ga = geo_args();
                    ug[G] = geo_uniform(ga, (grp + gstep) * G);
                    ug[G + 1] = geo_uniform(ga, (grp + gstep) * G + 1);
                } else { ug[G] = ug[0]; ug[G + 1] = ug[G == 1 ? 0 : 1]; }   // nothing follows: re-build this group's first footprint (never read)
            }
            if (fc) { wait_vmcnt<0>(); __syncthreads(); }
            // one block = one tile x one chunk: t and the accumulators are compile-time constants after inlining.
            // (c..): accumulators of column half 0 (NH = 1: of the tile), (d..): of column half 1 (NH = 2 only)
            auto run_block = [&](const int t, floatx16& c00, floatx16& c01, floatx16& c10, floatx16& c11,
                                 floatx16& d00, floatx16& d01, floatx16& d10, floatx16& d11) __attribute__((always_inline)) {
                // the footprint this block builds (for the block after it) and the one after that (whose windows it loads)
                const TGeo un = t + 1 < G ? ug[t + 1] : (last_chunk ? ug[G] : ug[0]);
                // (G == 1: the block after the next one is this tile's chunk ch + 2, or -- from the second-to-last chunk on -- the
                //  next group's tile, whose geometry ug[G] holds from the start of the group)
                const TGeo un2 = G == 1 ? (ch + 2 < nchunk ? ug[0] : ug[G])
                                        : (t + 2 < G ? ug[t + 2] : (last_chunk ? ug[t + 2] : ug[t + 2 - G]));
                const int nc0 = t + 1 < G ? c0 : (last_chunk ? 0 : c0 + F2_CH);
                // Per virtual step (tap, column half): 12 MFMAs on two row blocks x two column blocks, in three groups ordered
                // a.l * b.h, a.h * b.h, a.h * b.l.  Fragment reads: the A fragments of the next TAP (double-buffered) behind the
                // first group of the tap's last half, the next step's hi weights behind the second group (the hi registers are
                // free by then), its lo weights behind the third: 8 (NH = 1) / 6 (NH = 2) ds_read_b128 per 12 MFMAs.  The order
                // is PINNED (sched_barrier): left alone, hipcc sinks every ds_read to just in front of its MFMA.
                AFr a[2][2];
                BH bh, bl;
                unsigned cur[2];
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) { cur[rb] = (unsigned)lanepix[t][rb]; read_a(a[0][rb], cur[rb], vmask[t][rb], 0); }
                read_bh(bh, 0);
                read_bl(bl, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int tap = v / NH, half = v % NH;
                    const AFr& a0 = a[tap & 1][0];
                    const AFr& a1 = a[tap & 1][1];
                    if (EARLY_W && v == VB && t + 1 == G) {
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_barrier();
                        load_weights(nc0, 0, VB);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // (F32: the first fragment read, .h, holds k quads {0, 1} and pairs with the first weight plane, .l with the second:
                    //  groups (a0.h x bh), (a1.h x bh), (a.l x bl))
                    if (F32) { if (half == 0) mfma2(a0.h, bh, c00, c01); else mfma2(a0.h, bh, d00, d01); }
                    else if (half == 0) { mfma2(a0.l, bh, c00, c01); mfma2(a1.l, bh, c10, c11); }
                    else { mfma2(a0.l, bh, d00, d01); mfma2(a1.l, bh, d10, d11); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (half == NH - 1 && tap + 1 < NT) {
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) { cur[rb] = next_tap(cur[rb], tap); read_a(a[(tap + 1) & 1][rb], cur[rb], vmask[t][rb], tap + 1); }
                    }
                    if (v == 0) {
                        if (FUSED) { wx = settle(wpend); wpend = windows_of(un2.wb); }
                        fetch_block(un, nc0);
                    }
                    if (v == CS - 1) conv_consts();
                    __builtin_amdgcn_sched_barrier(0);
                    if (F32) { if (half == 0) mfma2(a1.h, bh, c10, c11); else mfma2(a1.h, bh, d10, d11); }
                    else if (half == 0) { mfma2(a0.h, bh, c00, c01); mfma2(a1.h, bh, c10, c11); }
                    else { mfma2(a0.h, bh, d00, d01); mfma2(a1.h, bh, d10, d11); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (RING && (v + 1) % TG == 0 && v + 1 < NV) {
                        // last step of tap group k - 1: the fragments of this step are in registers on every wave once all have
                        // passed the barrier, so the half that held group k - 1 is dead -- group k + 1 (or the next block's
                        // group 0) goes there; group k, whose first fragments are read right below, was issued TG steps ago
                        const int k = (v + 1) / TG;
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // group k landed; this step's lo fragments arrived
                        __builtin_amdgcn_s_barrier();
                        if (k + 1 < NG) load_weights(c0, (k + 1) * TG, (k + 2) * TG < NV ? (k + 2) * TG : NV);
                        else load_weights(nc0, 0, TG);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (v + 1 < NV) read_bh(bh, v + 1);
#pragma unroll
                    for (int q = 0; q < WS_NFV; ++q)
                        if (v == CS + q * CSTRIDE) convert_slice(q);
                    __builtin_amdgcn_sched_barrier(0);
                    if (F32) {
                        if (half == 0) { mfma2(a0.l, bl, c00, c01); mfma2(a1.l, bl, c10, c11); }
                        else { mfma2(a0.l, bl, d00, d01); mfma2(a1.l, bl, d10, d11); }
                    } else if (half == 0) { mfma2(a0.h, bl, c00, c01); mfma2(a1.h, bl, c10, c11); }
                    else { mfma2(a0.h, bl, d00, d01); mfma2(a1.h, bl, d10, d11); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (v + 1 < NV) read_bl(bl, v + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();                     // every wave has read its last fragments of this footprint (and of the weights)
                if (RING) load_weights(nc0, TG, 2 * TG);                // next block's group 1 into half 1 (group NG - 1 is done with it)
                else if (t + 1 == G) load_weights(nc0, EARLY_W ? VB : 0, NV);   // the (rest of the) next chunk's weights fly while the footprint is written
                write_footprint();
                // RING: the next block's group 0 (issued at the last refresh point) must have landed; its group 1 -- at least
                // 4 TG / 8 LDS-DMA pieces per wave, just issued, completing in order behind it -- is waited for at step TG - 1
                if (RING) wait_vmcnt<(4 * TG) / 8>();
                else if (t + 1 == G) wait_vmcnt<0>();
                __syncthreads();
            };
            if (NH == 2) run_block(0, acc000, acc001, acc010, acc011, acc100, acc101, acc110, acc111);
            else if (G == 1) run_block(0, acc000, acc001, acc010, acc011, acc000, acc001, acc010, acc011);      // RING: one tile per group
            else {
                run_block(0, acc000, acc001, acc010, acc011, acc000, acc001, acc010, acc011);
                run_block(1, acc100, acc101, acc110, acc111, acc100, acc101, acc110, acc111);
            }
        }
        // ---- group complete: epilogue parameters through the kernel-argument pointer
        {
            KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(q));
            EpiArgs e;
            e.bias = q->bias; e.ps = q->ps; e.pt = q->pt; e.res = q->res; e.out = q->out;
            e.M = q->M; e.Cout = q->Cout; e.act = q->act; e.pp = q->pp; e.poolkind = q->poolkind;
            e.out_np = q->out_hl ? q->out_np : 0u; e.out_f16 = q->out_f16;
            // t: NH = 1: tile of the group; NH = 2: column half of the group's one tile
            auto finish = [&](const int t, const int rb, floatx16& c0acc, floatx16& c1acc) __attribute__((always_inline)) {
                const int tile = NH == 2 ? grp : grp * G + t;
                const int nc = NH == 2 ? n0 + BN * t : n0;
                const long long row0 = (long long)tile * TMR + (wv * 2 + rb) * 32;
                if (RING) { const long long mend = (long long)tile * TMR + TMR; e.M = mend < (long long)M ? mend : (long long)M; }
                if (tile < ntiles) {
                    if (TR && EPI == 1) epilogue_tr<false, EpiArgs, true>(e, c0acc, c1acc, row0 + li, nc, lh);
                    else if (TR) epilogue_tr(e, c0acc, c1acc, row0 + li, nc, lh);
                    else if (EPI == 1) {
                        epilogue_pool_relu(e, c0acc, row0, nc + li, lh);
                        if (NCB == 2) epilogue_pool_relu(e, c1acc, row0, nc + 32 + li, lh);
                    } else {
                        epilogue_tile(e, c0acc, row0, nc + li, lh);
                        if (NCB == 2) epilogue_tile(e, c1acc, row0, nc + 32 + li, lh);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { c0acc[i] = 0.f; c1acc[i] = 0.f; }
            };
            finish(0, 0, acc000, acc001); finish(0, 1, acc010, acc011);
            if (NH == 2 || G == 2) { finish(1, 0, acc100, acc101); finish(1, 1, acc110, acc111); }
        }
        if (!last_group) {
            group_geometry(grp + gstep);
            // (ug[G], ug[G + 1] are rewritten at the next group's last chunk)
        }
    }
}

template <int KH, int KW>
void launch_ws_shape(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused) {
    if constexpr (KH * KW >= 8 && KH * KW <= WS_MAXNT) {
#define ISS_WS_LAUNCH(...) hipLaunchKernelGGL((conv_x3_ws_kernel<KH, KW, __VA_ARGS__>), grid, dim3(512), 0, st, a)
        // instantiated for the shared-first-layer convolution (the dominant launch of the segmenter nets) -- every
        // instantiation costs minutes of compile time; the other footprint layers stay on conv_x3_fp_kernel
        if (!fused) return;
        const bool fast = tr ? epi_is_simple_tr(a) : epi_is_pool_relu_any(a);
        if (padded) {
            if (tr) { if (fast) ISS_WS_LAUNCH(true, true, true, 1, 1); else ISS_WS_LAUNCH(true, true, true); }
            else { if (fast) ISS_WS_LAUNCH(true, false, true, 1, 1); else ISS_WS_LAUNCH(true, false, true); }
        } else {
            if (tr) { if (fast) ISS_WS_LAUNCH(false, true, true, 1, 1); else ISS_WS_LAUNCH(false, true, true); }
            else { if (fast) ISS_WS_LAUNCH(false, false, true, 1, 1); else ISS_WS_LAUNCH(false, false, true); }
        }
#undef ISS_WS_LAUNCH
    }
}

// Ring form (more than WS_MAXNT taps): first-layer-fused, row-major epilogue only (cnn_ws_e.hip)
inline bool iss_ws_ring_compiled(int kh, int kw) { return (kh == 7 && kw == 7) || (kh == 5 && kw == 5) || (kh == 4 && kw == 5); }
void iss_ws_launch_ring_7x7(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded);
void iss_ws_launch_ring_5x5(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded);      // cnn_ws_i.hip
void iss_ws_launch_ring_4x5(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded);      // cnn_ws_j.hip
inline void iss_ws_launch_ring(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) {
    if (a.H_k == 7) iss_ws_launch_ring_7x7(a, grid, st, padded);
    else if (a.H_k == 5) iss_ws_launch_ring_5x5(a, grid, st, padded);
    else iss_ws_launch_ring_4x5(a, grid, st, padded);
}
// FS form (zero-padded first layer in front): first-layer-fused, row-major epilogue only (cnn_ws_f.hip, cnn_ws_g.hip)
inline bool iss_ws_fs_compiled(int kh, int kw) { return (kh == 5 && kw == 3) || (kh == 3 && kw == 3); }
void iss_ws_launch_fs_5x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded);
void iss_ws_launch_fs_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded);
void iss_ws_launch_fs_5x3_tr(const ConvArgs& a, dim3 grid, hipStream_t st);   // unpadded, no fused pool: transposed simple epilogue (bias + relu)
template <int KH, int KW, bool FS_>
void launch_ws_fused_rowmajor(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) {
    const bool fast = epi_is_pool_relu_any(a);
#define ISS_WS_LAUNCH2(P_, E_) hipLaunchKernelGGL((conv_x3_ws_kernel<KH, KW, P_, false, true, 1, E_, FS_>), grid, dim3(512), 0, st, a)
    if (padded) { if (fast) ISS_WS_LAUNCH2(true, 1); else ISS_WS_LAUNCH2(true, 0); }
    else { if (fast) ISS_WS_LAUNCH2(false, 1); else ISS_WS_LAUNCH2(false, 0); }
#undef ISS_WS_LAUNCH2
}

// NCB = 1 form (cnn_ws_k.hip): first-layer-fused unpadded 5x3 with <= 32 output channels, pooled relu epilogue
inline bool iss_ws_ncb1_compiled(int kh, int kw) { return kh == 5 && kw == 3; }
void iss_ws_launch_ncb1_5x3(const ConvArgs& a, dim3 grid, hipStream_t st);
// exact-f32 form (cnn_ws_h.hip): the fused 5x3 layer (pooled relu epilogue) and the unpadded 3x3 NH = 2 layers (simple epilogues)
inline bool iss_ws_f32_fused_compiled(int kh, int kw) { return kh == 5 && kw == 3; }
void iss_ws_launch_f32_fused_5x3(const ConvArgs& a, dim3 grid, hipStream_t st);
void iss_ws_launch_f32_nh2_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr);

// Plain (not first-layer-fused) use of the kernel: zero-padded 3x3 stride-1 layers whose 128-row tile does not fit
// conv_x3_fp_kernel's 360-pixel footprint because the image is WIDE (ResNet-101's 32 -> 32 convolutions at 64 x 144: 580
// pixels per 128 rows, 802 per 512 rows <= WS_PIX) -- they ran on the gather kernel at 6 x their bandwidth bound.
inline bool iss_ws_plain_compiled(int kh, int kw) { return kh == 3 && kw == 3; }
void iss_ws_launch_plain_3x3(const ConvArgs& a, dim3 grid, hipStream_t st);     // padded, transposed epilogue (cnn_ws_c.hip)
void iss_ws_launch_plain_3x3_unpadded(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr);   // tr: bias + relu transposed; else pooled relu
// NH = 2 (see the kernel): unpadded 3x3 layers with a multiple of 128 output channels (cnn_ws_d.hip); tr: transposed epilogue
inline bool iss_ws_nh2_compiled(int kh, int kw) { return kh == 3 && kw == 3; }
void iss_ws_launch_nh2_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr);
void iss_ws_launch_nh2_3x3_padded(const ConvArgs& a, dim3 grid, hipStream_t st);    // transposed + simple epilogue (epi_is_simple_tr)
void iss_ws_launch_nh2_3x3_padded_pool(const ConvArgs& a, dim3 grid, hipStream_t st);   // row-major, relu + fused max-pool (epi_is_pool_relu)

}  // namespace issk

// Filter shapes the weight-stationary kernel is instantiated for (cnn_ws.hip)
#define ISS_WS_SHAPES_A(X) X(5, 3)
#define ISS_WS_SHAPES_B(X) X(3, 3)
#define ISS_WS_SHAPES(X) ISS_WS_SHAPES_A(X) ISS_WS_SHAPES_B(X)
#define ISS_WS_DECL(KH_, KW_) void iss_ws_launch_##KH_##x##KW_(const issk::ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused);
ISS_WS_SHAPES(ISS_WS_DECL)
#undef ISS_WS_DECL
