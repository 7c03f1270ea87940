// conv_x3_wq3_kernel: the one-wave-per-SIMD structure of conv_wq.h for the unpadded 3x3 layers with 128 output channels per
// workgroup (conv3 / conv4 of the segmenter nets: 64 -> 128 and 128 -> 128), which ran on conv_x3_ws_kernel<3,3,..,NH=2> at
// 50 / 58 % matrix-pipe occupancy with a serial epilogue (conv3's 2.6 GB of output per launch was a store tail).
//
// 256 threads = 4 waves, one per SIMD; a wave owns 64 rows x 128 channels of a tile of <= 256 rows (2 row blocks x 4 column
// blocks = 8 accumulators = 128 registers), two tiles per group = 256 accumulator registers.  Per tap 24 slots of ONE MFMA
// (three split-operand terms x 2 row blocks x 4 column blocks) with twelve fragment reads (4 A, 8 B), one per slot.
// LDS = [9 taps x 2 column halves x 4 KB of weights = 72 KB][two footprints of 512 pixels x 64 B = 64 KB]; footprint layout,
// running tap addresses, the single barrier per block, the weight refresh behind the last tap and the epilogue of the OTHER
// tile's accumulators behind the MFMAs of the next block are those of conv_wq.h (read its header first).  Differences:
//   * plain NHWC f32 input (no fused first layer): a slice is fetched with one clamped address and split into bf16 hi / lo;
//   * the row-parity key of the footprint swizzle is the parity of the FLATTENED input row (sample * H + y) -- the reader knows
//     it from its output row, the writer from (first pixel's column + pixel offset) / W;
//   * KIND 0 (conv3: bias + relu, no pool): MFMAs issued transposed (C^T = W A^T), a lane holds 4 consecutive channels of one
//     pixel -> one 16-byte store per (accumulator, register group), bias fetched per unit as a float4;
//     KIND 1 (conv4: relu + 2 x 1 max-pool): rows in registers, a pool window is two consecutive registers -> two dword stores
//     per unit (32 lanes x 4 B = one full 128-byte line each).
#pragma once
#include "conv_wq.h"

namespace issk {

constexpr int WQ3_PIX = 512;                       // footprint capacity in pixels (host-validated per launch)
constexpr int WQ3_FB = WQ3_PIX * WQ_ROW;           // 32 KB per footprint
constexpr int WQ3_NFV = WQ3_PIX / 64;              // 8 slices of 64 pixels: 256 threads x 4 channels each
constexpr int WQ3_TM = 256;                        // rows per tile at most: 4 waves x 64
constexpr int WQ3_NV = 18;                         // resident 4 KB weight tiles: 9 taps x 2 column halves
constexpr int wq3_lds_bytes() { return WQ3_NV * F2_BST + 2 * WQ3_FB + 512; }     // + the 128 bias values of the workgroup's columns

#ifndef ISS_WQ3_EXP
#define ISS_WQ3_EXP 0
#endif

template <int KIND>
__global__ __launch_bounds__(256, 1) void conv_x3_wq3_kernel(const ConvArgs p) {
    constexpr bool X_NOEPI = ISS_WQ3_EXP & 4, X_NOCHUNKBAR = ISS_WQ3_EXP & 64, X_NODMA = ISS_WQ3_EXP & 128, X_NOBAR = ISS_WQ3_EXP & 2;
    constexpr bool X_NOFETCH = ISS_WQ3_EXP & 1, X_NOCONV = ISS_WQ3_EXP & 8;      // bit 0: no footprint loads, bit 3: no conversion / LDS stores
    constexpr int KH = 3, KW = 3, NT = 9, G = 2;
    constexpr bool TR = KIND == 0;
    static_assert(wq3_lds_bytes() <= 160 * 1024 && (WQ3_NV * F2_BST) % 4096 == 0 && WQ3_FB % 2048 == 0, "");
    __shared__ __attribute__((aligned(4096))) unsigned char smem[wq3_lds_bytes()];     // [18 weight tiles][footprint 0][footprint 1]
    const unsigned sB_base = (unsigned)(size_t)smem;
    const unsigned sF0 = sB_base + WQ3_NV * F2_BST;
    const unsigned sBias = sF0 + 2 * WQ3_FB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);          // 0..3
    const int n0 = blockIdx.y * (2 * BN);
    const int li = lane & 31, lh = lane >> 5;
    const int M = (int)p.M;
    const int TMR = p.tmr;                           // rows per tile (<= 256, multiple of 4)
    const int ntiles = (M + TMR - 1) / TMR;
    const int ngroups = (ntiles + G - 1) / G;
    int grp = (int)blockIdx.x;
    if (grp >= ngroups) return;
    int totpix;                                      // samples * H * W
    { const int spp = p.Hq * p.Wq * p.pp; totpix = (int)(p.img_stride / p.Cin) * (M / spp); }

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
    struct TGeo { int p_lo, need, xlo, rpar; };      // uniform per tile: first pixel, pixels needed, its column, parity of its flattened row
    auto clamp_tile = [&](int t) { return t < ntiles ? t : ntiles - 1; };
    auto geo_uniform = [&](const GeoArgs& ga, int tile) {
        TGeo u;
        const int m0 = clamp_tile(tile) * TMR;
        int b, oy, ox;
        map_row32(ga, m0, b, oy, ox);
        u.p_lo = (b * ga.H + oy) * ga.W + ox;
        u.xlo = ox; u.rpar = (b * ga.H + oy) & 1;
        const int ml = m0 + TMR - 1 < M - 1 ? m0 + TMR - 1 : M - 1;
        int b2, oy2, ox2;
        map_row32(ga, ml, b2, oy2, ox2);
        u.need = (b2 * ga.H + (oy2 + KH - 1)) * ga.W + (ox2 + KW - 1) - u.p_lo + 1;
        return u;
    };
    // LDS byte address of the lane's first tap in footprint `fb` for EVEN filter rows (32-byte half = k-half ^ row parity)
    auto geo_lane = [&](const GeoArgs& ga, int tile, int rb, const TGeo& u, int fb) {
        const int m0 = clamp_tile(tile) * TMR;
        const int m = m0 + (wv * 2 + rb) * 32 + li;
        int b, oy, ox;
        map_row32(ga, m < M ? m : m0, b, oy, ox);
        const int lp = (b * ga.H + oy) * ga.W + ox - u.p_lo;
        const int hi = WQ3_PIX - 1 - ((KH - 1) * ga.W + (KW - 1));
        return sF0 + (unsigned)(fb * WQ3_FB) + (unsigned)((lp < 0 ? 0 : (lp > hi ? hi : lp)) * WQ_ROW) + (unsigned)(((lh ^ (b * ga.H + oy)) & 1) << 5);
    };

    // ---- weights of one 16-channel chunk: 18 tiles of 4 KB (tile k = tap k >> 1, column half k & 1) = 72 pieces of 1 KB; wave w
    // moves pieces w, w + 4, ...: (row half, plane) = (w & 1, w >> 1) for every one of its pieces, tile = piece >> 2
    const int w_half = wv & 1, w_plane = (wv >> 1) & 1;
    unsigned boff_w[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int n = 32 * w_half + (lane >> 1), h = (lane & 1) ^ ((n >> 3) & 1);
        const int row = n0 + 64 * ch + n;
        boff_w[ch] = 2u * ((unsigned)(row < p.Cout ? row : 0) * (unsigned)p.Kpad + (unsigned)(h * 8));      // bytes
    }
    unsigned wdst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_base + w_plane * 2048 + w_half * 1024));     // the wave's piece in tile 0
    auto load_weight_tile = [&](int c0, int k) {     // k: compile-time
        const uint16_t* src = (w_plane ? p.wl : p.wh) + ((k >> 1) * p.Cin + c0);
        asm volatile("" : "+s"(wdst0));              // (one scalar add per piece instead of 18 hoisted destinations)
        glds16(src, boff_w[k & 1], wdst0 + (unsigned)(k * F2_BST));
    };
    unsigned bread = sB_base + (unsigned)((2 * li + (lh ^ ((li >> 3) & 1))) * 16);
    asm volatile("" : "+v"(bread));
    // B fragment of (tap, column block cb): tile (tap * 2 + (cb >> 1)), 32-column half cb & 1; hi plane at + 0, lo at + 2048
    auto b_off = [&](int tap, int cb) { return (unsigned)((tap * 2 + (cb >> 1)) * F2_BST + (cb & 1) * 1024); };

    // ---- footprint slices: thread -> pixel 64 q + (tid >> 2), channels [c0 + 4 (tid & 3), + 4)
    const int cg = tid & 3, prow = tid >> 2;
    float4 fv[WQ3_NFV];
    unsigned fm[WQ3_NFV];                            // per slice: row parity of the thread's pixel
    const int magicW = (65536 + p.W - 1) / p.W;      // x / W == (x * magicW) >> 16 for x < WQ3_PIX + W (host-checked)
    const unsigned cin4 = (unsigned)p.Cin * 4u, cg16 = (unsigned)cg * 16u;
    auto fetch_1 = [&](const TGeo& u, int q) {       // q: compile-time.  Row parity of the pixel
        int d = u.xlo + prow + 64 * q;
        asm volatile("" : "+v"(d));
        fm[q] = (unsigned)(u.rpar + ((d * magicW) >> 16)) & 1u;
    };
    auto fetch_2 = [&](const TGeo& u, int c0, int q) {        // clamped address + load
        const int qq = 64 * q < u.need ? q : 0;      // unneeded slices re-load slice 0
        int gp = u.p_lo + prow + 64 * qq;
        gp = gp < 0 ? 0 : (gp > totpix - 1 ? totpix - 1 : gp);
        fv[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.in + c0) + ((unsigned)gp * cin4 + cg16));
    };
    const unsigned wofs = (unsigned)(prow * WQ_ROW + (cg >> 1) * 32 + (cg & 1) * 8);
    const int wsgn = (cg >> 1) ? -32 : 32;
    struct Cv { float4 v; bf16x4 h; };
    auto convert_1 = [&](Cv& c, int q) {             // hi parts
        float4 v = fv[q];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
        c.v = v;
        c.h[0] = (__bf16)v.x; c.h[1] = (__bf16)v.y; c.h[2] = (__bf16)v.z; c.h[3] = (__bf16)v.w;
    };
    auto convert_2 = [&](Cv& c) {                    // residuals, first half
        c.v.x = c.v.x - (float)c.h[0]; c.v.y = c.v.y - (float)c.h[1];
    };
    auto convert_3 = [&](Cv& c) {
        c.v.z = c.v.z - (float)c.h[2]; c.v.w = c.v.w - (float)c.h[3];
    };
    auto convert_4 = [&](Cv& c, int q, unsigned wbase) {      // lo parts + the two 8-byte stores
        bf16x4 l;
        l[0] = (__bf16)c.v.x; l[1] = (__bf16)c.v.y; l[2] = (__bf16)c.v.z; l[3] = (__bf16)c.v.w;
        const unsigned a = wbase + (unsigned)(wsgn * (int)fm[q]);
        *(LdsW8)(a + (unsigned)(q * 64 * WQ_ROW)) = c.h;
        *(LdsW8)(a + (unsigned)(q * 64 * WQ_ROW + 16)) = l;
    };

    struct AFr { bf16x8 h, l; };
    const unsigned wstep2 = (unsigned)(2 * p.W * WQ_ROW);
    auto mfma = [&](const bf16x8& a, const bf16x8& b, const floatx16& c) {
        if (TR) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };

    // accumulators: acc<tile><row block><column block>
    floatx16 c000, c001, c002, c003, c010, c011, c012, c013, c100, c101, c102, c103, c110, c111, c112, c113;
    {
        floatx16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        c000 = z; c001 = z; c002 = z; c003 = z; c010 = z; c011 = z; c012 = z; c013 = z;
        c100 = z; c101 = z; c102 = z; c103 = z; c110 = z; c111 = z; c112 = z; c113 = z;
    }

    // ---- epilogue pieces (buffer stores: an offset beyond the tensor is dropped by the hardware)
    struct Epi { const float* bias; float* out; int cout; };
    Epi ep;
    {
        KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(q));
        ep.bias = q->bias; ep.out = q->out; ep.cout = q->Cout;
    }
    const unsigned out_bytes = TR ? (unsigned)M * (unsigned)ep.cout * 4u : (unsigned)(M >> 1) * (unsigned)ep.cout * 4u;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, (int)out_bytes, 0x00020000);
    constexpr unsigned E_INVALID = 0xFFFF0000u;      // (the host keeps the output below 0xFFF00000 bytes)
    int rowb = ep.cout * 4;                          // bytes per output row
    // KIND 1: bias of the lane's columns n0 + 32 cb + li
    float ebias[4] = {0.f, 0.f, 0.f, 0.f};
    if (!TR) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) ebias[cb] = ep.bias[n0 + 32 * cb + li < ep.cout ? n0 + 32 * cb + li : 0];
    }
    if (TR && tid < 128) *(LdsW4)(sBias + (unsigned)(tid * 4)) = __float_as_uint(ep.bias[n0 + tid < ep.cout ? n0 + tid : 0]);     // (visible after the prologue's barrier)
    const int wrow = TR ? wv * 64 + li : wv * 64 + 4 * lh;     // the lane's first row inside the tile (row block 0, group 0)
    float4 e_bb[2], e_v = make_float4(0.f, 0.f, 0.f, 0.f);
    e_bb[0] = e_v; e_bb[1] = e_v;
    float e_p0 = 0.f, e_p1 = 0.f;
    // KIND 0, unit (rb, cb, g): channels n0 + 32 cb + 8 g + 4 lh + {0..3} of pixel row (wv * 2 + rb) * 32 + li
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const f32x4 __attribute__((address_space(3)))* LdsRF4;
    const unsigned bias_rd = sBias + (unsigned)(lh * 16);
    auto epi0_a = [&](int unit) {                    // the bias of unit `unit` (channels 32 cb + 8 g + 4 lh + {0..3}) into set unit & 1:
        const int cb = (unit >> 2) & 3, g = unit & 3;          // read one unit ahead of its use
        const f32x4 t = *(LdsRF4)(bias_rd + (unsigned)((32 * cb + 8 * g) * 4));
        e_bb[unit & 1] = make_float4(t[0], t[1], t[2], t[3]);
    };
    auto epi0_b = [&](const floatx16& acc, int unit) {         // + bias, relu
        const int g = unit & 3;
        const float4 e_b = e_bb[unit & 1];
        e_v = make_float4(fmaxf(acc[4 * g] + e_b.x, 0.f), fmaxf(acc[4 * g + 1] + e_b.y, 0.f),
                          fmaxf(acc[4 * g + 2] + e_b.z, 0.f), fmaxf(acc[4 * g + 3] + e_b.w, 0.f));
        asm volatile("" : "+v"(e_v.x), "+v"(e_v.y), "+v"(e_v.z), "+v"(e_v.w));
    };
    // vb: byte offset of (row tile * tmr + wv * 64 + li, channel n0 + 4 lh) in `out`
    auto epi0_c = [&](int rb, int cb, int g, unsigned vb, int tile_rows) {
        int wr = wrow;
        asm volatile("" : "+v"(wr), "+s"(rowb));
        const bool ok = wr < tile_rows - rb * 32;
        const unsigned off = ok ? vb : E_INVALID;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 d;
        d[0] = __float_as_uint(e_v.x); d[1] = __float_as_uint(e_v.y); d[2] = __float_as_uint(e_v.z); d[3] = __float_as_uint(e_v.w);
        if (!X_NOEPI) __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, (int)off, rb * 32 * rowb + (32 * cb + 8 * g) * 4, 0);
    };
    // KIND 1, unit (rb, cb, g): rows 8 g + 4 lh + {0..3} of the row block = two 2 x 1 pool windows, column n0 + 32 cb + li
    auto epi1_a = [&](const floatx16& acc, int cb, int g) {
        e_p0 = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]) + ebias[cb], 0.f);
        e_p1 = fmaxf(fmaxf(acc[4 * g + 2], acc[4 * g + 3]) + ebias[cb], 0.f);
        asm volatile("" : "+v"(e_p0), "+v"(e_p1));
    };
    // vb: byte offset of (pooled row (tile * tmr + wv * 64) / 2 + 2 lh, column n0 + li) in `out`
    auto epi1_b = [&](int rb, int cb, int g, unsigned vb, int tile_rows) {
        int wr = wrow;
        asm volatile("" : "+v"(wr), "+s"(rowb));
        const bool ok = wr < tile_rows - (rb * 32 + 8 * g);
        const unsigned off = ok ? vb : E_INVALID;
        if (!X_NOEPI) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e_p0), orsrc, (int)off, (rb * 16 + 4 * g) * rowb + cb * 128, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(e_p1), orsrc, (int)off, (rb * 16 + 4 * g + 1) * rowb + cb * 128, 0);
        }
    };
    auto epi_base = [&](int tile) {
        if (TR) return (unsigned)(((tile * TMR + wv * 64 + li) * ep.cout + n0 + 4 * lh) * 4);
        return (unsigned)(((tile * (TMR >> 1) + wv * 32 + 2 * lh) * ep.cout + n0 + li) * 4);
    };
    auto tile_rows_of = [&](int tile) { const int r = M - tile * TMR; return tile < ntiles ? (r < TMR ? r : TMR) : 0; };

    // ---- prologue
    TGeo ug[G + 2];
    unsigned lb[G][2];
    unsigned lbn[2];
    auto group_geometry = [&](int g0) {
        const GeoArgs ga = geo_args();
#pragma unroll
        for (int t = 0; t < G; ++t) {
            ug[t] = geo_uniform(ga, g0 * G + t);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) lb[t][rb] = geo_lane(ga, g0 * G + t, rb, ug[t], t);
        }
    };
    group_geometry(grp);
    ug[G] = ug[0]; ug[G + 1] = ug[1];
    lbn[0] = lb[0][0]; lbn[1] = lb[0][1];
    Cv cva, cvb;
#pragma unroll
    for (int q = 0; q < WQ3_NFV; ++q) { fetch_1(ug[0], q); fetch_2(ug[0], 0, q); }
#pragma unroll
    for (int q = 0; q < WQ3_NFV; ++q) { convert_1(cva, q); convert_2(cva); convert_3(cva); convert_4(cva, q, sF0 + wofs); }
#pragma unroll
    for (int k = 0; k < WQ3_NV; ++k) load_weight_tile(0, k);
    wait_vmcnt<0>();
    __syncthreads();

    AFr a[2][2];
    bf16x8 bh[2][4], bl[4], blast[4];
    unsigned re[2], ro[2];
    auto read_a_h = [&](AFr& f, unsigned ad, int kx) { f.h = *(LdsR16)(ad + (unsigned)(kx * WQ_ROW)); };
    auto read_a_l = [&](AFr& f, unsigned ad, int kx) { f.l = *(LdsR16)(ad + (unsigned)(kx * WQ_ROW + 16)); };
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        re[rb] = lb[0][rb]; ro[rb] = (lb[0][rb] ^ 32u) + (unsigned)(p.W * WQ_ROW);
        read_a_h(a[0][rb], re[rb], 0); read_a_l(a[0][rb], re[rb], 0);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bh[0][cb] = *(LdsR16)(bread + b_off(0, cb));

    const int nchunk = p.Cin / F2_CH;                // >= 2 (host-checked)
    const int gstep = (int)gridDim.x;
    int prev_tile1 = ntiles;
    for (; grp < ngroups; grp += gstep) {
        const bool last_group = grp + gstep >= ngroups;
        auto run_block = [&](const int t, const bool ZC, const bool EP, const int c0, const bool last_chunk,
                             floatx16& d00, floatx16& d01, floatx16& d02, floatx16& d03,
                             floatx16& d10, floatx16& d11, floatx16& d12, floatx16& d13,
                             const floatx16& o00, const floatx16& o01, const floatx16& o02, const floatx16& o03,
                             const floatx16& o10, const floatx16& o11, const floatx16& o12, const floatx16& o13,
                             const int etile) __attribute__((always_inline)) {
            const TGeo un = t == 0 ? ug[1] : (last_chunk ? ug[G] : ug[0]);
            const int nc0 = t == 0 ? c0 : (last_chunk ? 0 : c0 + F2_CH);
            const unsigned wbase = sF0 + (unsigned)((1 - t) * WQ3_FB) + wofs;
            unsigned vb = E_INVALID;
            int erows = 0;
            if (EP) { vb = epi_base(etile); erows = tile_rows_of(etile); }
#pragma unroll
            for (int v = 0; v < NT; ++v) {
                const int cs = (v + t) & 1, ns = cs ^ 1;
                const bool last = v + 1 == NT;
                const int ky1 = (v + 1) / KW, kx1 = (v + 1) % KW;
                if (last && !X_NOBAR) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                // 24 slots of ONE MFMA: s = term * 8 + rb * 4 + cb, terms a.l x b.h, a.h x b.h, a.h x b.l
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    const int term = s >> 3, rb = (s >> 2) & 1, cb = s & 3;
                    floatx16& e = rb == 0 ? (cb == 0 ? d00 : cb == 1 ? d01 : cb == 2 ? d02 : d03) : (cb == 0 ? d10 : cb == 1 ? d11 : cb == 2 ? d12 : d13);
                    const bf16x8& av = term == 0 ? a[cs][rb].l : a[cs][rb].h;
                    const bf16x8& bv = term == 2 ? (last ? blast[cb] : bl[cb]) : bh[cs][cb];
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
                    // ---- slots 0..11: fragment reads, one per slot.  0..3: the lo weights of THIS tap (used from slot 16 on);
                    // 4..7: the A fragments of the next tap (a.l x 2, a.h x 2); 8..11: its hi weights
                    if (s < 12) {
                        const bool bok = !last || t == 0;            // (t == 1: the next chunk's weights are still in flight)
                        const int tn = last ? 0 : v + 1;
                        if (!last) {
                            if (s == 4 && kx1 == 0 && ky1 >= 2) {
#pragma unroll
                                for (int r = 0; r < 2; ++r) {
                                    if (ky1 & 1) { asm volatile("" : "+v"(ro[r])); ro[r] += wstep2; }
                                    else { asm volatile("" : "+v"(re[r])); re[r] += wstep2; }
                                }
                            }
                        } else if (s == 4) {
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                const unsigned nbr = t == 0 ? lb[1][r] : (last_chunk ? lbn[r] : lb[0][r]);
                                re[r] = nbr; ro[r] = (nbr ^ 32u) + (unsigned)(p.W * WQ_ROW);
                            }
                        }
                        const int ky = last ? 0 : ky1, kx = last ? 0 : kx1;
#define ISS_WQ_RA(r) ((ky & 1) ? ro[r] : re[r])
                        if (s < 4) { if (!last) bl[s] = *(LdsR16)(bread + b_off(v, s) + 2048); }
                        else if (s < 6) read_a_l(a[ns][s - 4], ISS_WQ_RA(s - 4), kx);
                        else if (s < 8) read_a_h(a[ns][s - 6], ISS_WQ_RA(s - 6), kx);
                        else if (bok) bh[ns][s - 8] = *(LdsR16)(bread + b_off(tn, s - 8));
#undef ISS_WQ_RA
                    }
                    // the last tap's lo weights, in the step before it (slots 12..15)
                    if (v + 2 == NT && s >= 12 && s < 16) blast[s - 12] = *(LdsR16)(bread + b_off(NT - 1, s - 12) + 2048);
                    // ---- the last step of a t == 1 block: the next chunk's weights, one LDS-DMA per slot (slots 0..17)
                    if (last && t == 1 && s < WQ3_NV && !X_NODMA) load_weight_tile(nc0, s);
                    // ---- footprint pipeline: slices 2 k, 2 k + 1 are fetched in step k (two pieces each, slots 20..23) and converted
                    // in step k + 3 (four pieces each, slots 12..15 and 16..19), k = 0..3
                    if (v <= 3 && s >= 20 && !X_NOFETCH) {
                        const int q = 2 * v + ((s - 20) >> 1);
                        if (((s - 20) & 1) == 0) fetch_1(un, q); else fetch_2(un, nc0, q);
                    }
                    if (X_NOCONV && !X_NOFETCH && v == 6 && s == 12) {
#pragma unroll
                        for (int q = 0; q < WQ3_NFV; ++q) asm volatile("" :: "v"(fv[q].x), "v"(fv[q].y), "v"(fv[q].z), "v"(fv[q].w));
                    }
                    if (v >= 3 && v <= 6 && s >= 12 && s < 20 && !X_NOCONV) {
                        const int q = 2 * (v - 3) + ((s - 12) >> 2);
                        Cv& c = s < 16 ? cva : cvb;
                        const int piece = (s - 12) & 3;
                        if (piece == 0) convert_1(c, q);
                        if (piece == 1) convert_2(c);
                        if (piece == 2) convert_3(c);
                        if (piece == 3) convert_4(c, q, wbase);
                    }
                    // ---- epilogue of the other accumulator set: 32 units (rb, cb, g); KIND 0: three pieces per unit in the read slots
                    // of steps 0..7 (four units per step); KIND 1: two pieces per unit in slots 0..7 of steps 0..7
                    if (EP && v <= 7) {
                        if (TR && s < 12) {
                            const int unit = v * 4 + s / 3;
                            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                            if (s % 3 == 0) { if (unit == 0) epi0_a(0); if (unit + 1 < 32) epi0_a(unit + 1); }
                            else if (s % 3 == 1) epi0_b(oa, unit);
                            else epi0_c(erb, ecb, eg, vb, erows);
                        }
                        if (!TR && s < 8) {
                            const int unit = v * 4 + s / 2;
                            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                            if (s % 2 == 0) epi1_a(oa, ecb, eg); else epi1_b(erb, ecb, eg, vb, erows);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t == 1) {                            // chunk boundary: the weights must have landed before anybody reads them
                if (!X_NOCHUNKBAR && !X_NOBAR) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) bh[(NT + t) & 1][cb] = *(LdsR16)(bread + b_off(0, cb));
            }
        };
#define ISS_WQ3_SET0 c000, c001, c002, c003, c010, c011, c012, c013
#define ISS_WQ3_SET1 c100, c101, c102, c103, c110, c111, c112, c113
        run_block(0, true, true, 0, false, ISS_WQ3_SET0, ISS_WQ3_SET1, prev_tile1);
        run_block(1, true, false, 0, false, ISS_WQ3_SET1, ISS_WQ3_SET0, 0);
        for (int ch = 1; ch + 1 < nchunk; ++ch) {
            run_block(0, false, false, ch * F2_CH, false, ISS_WQ3_SET0, ISS_WQ3_SET1, 0);
            run_block(1, false, false, ch * F2_CH, false, ISS_WQ3_SET1, ISS_WQ3_SET0, 0);
        }
        {
            if (!last_group) {
                const GeoArgs ga = geo_args();
                ug[G] = geo_uniform(ga, (grp + gstep) * G);
                ug[G + 1] = geo_uniform(ga, (grp + gstep) * G + 1);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) lbn[rb] = geo_lane(ga, (grp + gstep) * G, rb, ug[G], 0);
            } else {
                ug[G] = ug[0]; ug[G + 1] = ug[1];
                lbn[0] = lb[0][0]; lbn[1] = lb[0][1];
            }
            run_block(0, false, false, (nchunk - 1) * F2_CH, true, ISS_WQ3_SET0, ISS_WQ3_SET1, 0);
            run_block(1, false, true, (nchunk - 1) * F2_CH, true, ISS_WQ3_SET1, ISS_WQ3_SET0, grp * G);
        }
        prev_tile1 = grp * G + 1;
        if (!last_group) {
            lb[0][0] = lbn[0]; lb[0][1] = lbn[1];
            ug[0] = ug[G]; ug[1] = ug[G + 1];
            const GeoArgs ga = geo_args();
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) lb[1][rb] = geo_lane(ga, (grp + gstep) * G + 1, rb, ug[1], 1);
        }
    }
    // ---- the last group's tile 1: the only serial epilogue of the workgroup
    {
        const unsigned vb = epi_base(prev_tile1);
        const int erows = tile_rows_of(prev_tile1);
#pragma unroll
        for (int unit = 0; unit < 32; ++unit) {
            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
            const floatx16& oa = erb == 0 ? (ecb == 0 ? c100 : ecb == 1 ? c101 : ecb == 2 ? c102 : c103) : (ecb == 0 ? c110 : ecb == 1 ? c111 : ecb == 2 ? c112 : c113);
            if (TR) { epi0_a(unit); epi0_b(oa, unit); epi0_c(erb, ecb, eg, vb, erows); }
            else { epi1_a(oa, ecb, eg); epi1_b(erb, ecb, eg, vb, erows); }
        }
    }
#undef ISS_WQ3_SET0
#undef ISS_WQ3_SET1
}

void iss_wq3_launch(const ConvArgs& a, dim3 grid, hipStream_t st, int kind);

}  // namespace issk
