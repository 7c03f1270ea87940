// conv_x3_wq3h_kernel: conv_x3_wq3_kernel (conv_wq3.h: one wave per SIMD, unpadded 3x3, 128 output channels per workgroup, two
// 256-row tiles per group) on a CHL input (conv_common.h) -- round 6.
//
// Timing-only builds of conv_x3_wq3_kernel (profiles/r06_wq3_experiments.txt) price its input path -- global loads into registers,
// the bf16 hi / lo split (27 VALU per float4) and the LDS stores -- at 17-20 % of the launch, and its weight refresh (18 LDS-DMA
// pieces per wave in ONE tap, then vmcnt(0) + barrier) at another 10-18 %.  Here:
//   * the producer has already split the activations (conv_wq.h OUT_HL, or this kernel's own OUT_HL epilogue): a footprint is
//     four planes [k-half][hi | lo] of 512 pixels x 16 bytes, and plane w arrives by eight LDS-DMA instructions of wave w
//     (64 consecutive pixels = 1 KB of contiguous memory each): no VGPR, no VALU, no ds_write, no per-pixel address arithmetic.
//     A lane's tap (ky, kx) is at (lane address of the filter row) + 16 kx, its lo part 8 KB further: immediates;
//   * weights: 11 resident tap slots of 8 KB instead of 9.  Taps 0..6 of the next chunk overwrite this chunk's, tap v one tap
//     after every wave has read it for the last time (the waves publish "read tap v of pass n" in an LDS word each and the
//     loader checks the four words: no barrier); taps 7, 8 alternate between slots {7, 8} and {9, 10}, so the next chunk's are
//     fetched while this chunk's are still in use.  All 18 pieces of a wave are issued one tap pair per tap BEFORE the barrier of
//     the chunk's second block, which waits for them: the 18-piece burst and the chunk-boundary barrier are gone;
//   * OUT_HL: the epilogue splits its own output and writes CHL for the next 3x3 layer (kind 0: conv3 -> conv4).
// MFMA order per accumulator = conv_x3_wq3_kernel's (chunk, tap, term): results are bit-identical to the f32-input kernel's.
#pragma once
#include "conv_wq3.h"

namespace issk {

constexpr int WQH_PL = WQ3_PIX * 16;               // bytes per footprint plane (8 KB)
constexpr int WQH_FB = 4 * WQH_PL;                 // 32 KB per footprint
constexpr int WQH_NSLOT = 11;                      // resident tap slots (8 KB = two 64-column weight tiles each)
constexpr int WQH_SLOT = 2 * F2_BST;
constexpr int wqh_lds_bytes() { return WQH_NSLOT * WQH_SLOT + 2 * WQH_FB + 512 + 256; }     // + bias table + the waves' progress words

#ifndef ISS_WQH_EXP
#define ISS_WQH_EXP 0
#endif

template <int KIND, bool OUT_HL, bool F16 = false>
__global__ __launch_bounds__(256, 1) void conv_x3_wq3h_kernel(const ConvArgs p) {
    constexpr bool X_NOEPI = ISS_WQH_EXP & 4, X_NODMA = ISS_WQH_EXP & 128, X_NOADMA = ISS_WQH_EXP & 1, X_NOBAR = ISS_WQH_EXP & 2, X_NOFLAG = ISS_WQH_EXP & 16;      // timing-only experiment builds
    constexpr int KH = 3, KW = 3, NT = 9, G = 2;
    constexpr bool TR = KIND == 0;
    // OUT_HL: kind 0 writes CHL for the next 3x3 layer; kind 1 writes the pixel-major split layout of a DENSE layer behind the flatten
    // (conv_dhl.h)
    static_assert(wqh_lds_bytes() <= 160 * 1024, "");
    __shared__ __attribute__((aligned(4096))) unsigned char smem[wqh_lds_bytes()];     // [11 tap slots][footprint 0][footprint 1][bias][progress]
    const unsigned sB_base = (unsigned)(size_t)smem;
    const unsigned sF0 = sB_base + WQH_NSLOT * WQH_SLOT;
    const unsigned sBias = sF0 + 2 * WQH_FB;
    const unsigned sProg = sBias + 512;              // progress words: tap v (0..6) at + 16 v, one dword per wave

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
    auto clamp_tile = [&](int t) { return t < ntiles ? t : ntiles - 1; };
    // first input pixel of a tile (uniform)
    auto geo_plo = [&](const GeoArgs& ga, int tile) {
        int b, oy, ox;
        map_row32(ga, clamp_tile(tile) * TMR, b, oy, ox);
        return (b * ga.H + oy) * ga.W + ox;
    };
    // LDS byte address of the lane's first tap (hi part) in footprint `fb`: plane 2 lh, pixel = the lane's row's first input pixel
    auto geo_lane = [&](const GeoArgs& ga, int tile, int rb, int p_lo, int fb) {
        const int m0 = clamp_tile(tile) * TMR;
        const int m = m0 + (wv * 2 + rb) * 32 + li;
        int b, oy, ox;
        map_row32(ga, m < M ? m : m0, b, oy, ox);
        const int lp = (b * ga.H + oy) * ga.W + ox - p_lo;
        const int hi = WQ3_PIX - 1 - ((KH - 1) * ga.W + (KW - 1));
        return sF0 + (unsigned)(fb * WQH_FB) + (unsigned)(2 * lh * WQH_PL) + (unsigned)((lp < 0 ? 0 : (lp > hi ? hi : lp)) * 16);
    };

    // ---- weights of one 16-channel chunk: tap v = two 4 KB tiles (column halves) = one 8 KB slot = 8 pieces of 1 KB; wave w moves
    // pieces (row half, plane) = (w & 1, w >> 1) of both tiles of every tap
    const int w_half = wv & 1, w_plane = (wv >> 1) & 1;
    unsigned boff_w[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int n = 32 * w_half + (lane >> 1), h = (lane & 1) ^ ((n >> 3) & 1);
        const int row = n0 + 64 * ch + n;
        boff_w[ch] = 2u * ((unsigned)(row < p.Cout ? row : 0) * (unsigned)p.Kpad + (unsigned)(h * 8));      // bytes
    }
    unsigned wdst0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_base + w_plane * 2048 + w_half * 1024));     // the wave's piece in slot 0, tile 0
    // slot of tap v in chunk pass `par` (parity): taps 0..6 -> slots 0..6; taps 7, 8 -> slots 7, 8 (even passes) or 9, 10 (odd)
    auto load_weight_piece = [&](int c0, int v, int half, unsigned sw78) {     // v, half: compile-time; sw78: 0 or 2 * WQH_SLOT (uniform)
        const uint16_t* src = (w_plane ? p.wl : p.wh) + (v * p.Cin + c0);
        asm volatile("" : "+s"(wdst0));
        if (!X_NODMA) glds16_m0(src, boff_w[half], wdst0 + (unsigned)(v * WQH_SLOT + half * F2_BST) + (v >= 7 ? sw78 : 0u));
    };
    unsigned bread = sB_base + (unsigned)((2 * li + (lh ^ ((li >> 3) & 1))) * 16);
    unsigned bread78 = bread;                        // the same for taps 7, 8 of the current pass
    asm volatile("" : "+v"(bread), "+v"(bread78));
    // B fragment of (tap, column block cb): slot of the tap, tile cb >> 1, 32-column half cb & 1; hi plane at + 0, lo at + 2048
    auto b_addr = [&](int tap, int cb) { return (tap >= 7 ? bread78 : bread) + (unsigned)(tap * WQH_SLOT + (cb >> 1) * F2_BST + (cb & 1) * 1024); };

    // ---- footprint of a tile: plane wv by this wave, slices of 64 pixels
    const uint16_t* in_hl = reinterpret_cast<const uint16_t*>(p.in);
    const unsigned np16 = p.in_np * 16u;             // bytes per plane
    auto load_slice = [&](int p_lo, int c0, int q, int fb) {          // q: compile-time
        // plane (c0 / 16) * 4 + wv of the tensor, pixels p_lo + 64 q + lane
        const unsigned char* src = reinterpret_cast<const unsigned char*>(in_hl) + (size_t)((unsigned)(c0 >> 4) * 4u + (unsigned)wv) * np16;
        const unsigned off = (unsigned)(p_lo + 64 * q + lane) * 16u;
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sF0 + (unsigned)(fb * WQH_FB) + (unsigned)(wv * WQH_PL + q * 1024)));
        if (!X_NOADMA) glds16_m0(src, off, dst);
    };

    struct AFr { bf16x8 h, l; };
    const unsigned wstep = (unsigned)(p.W * 16);     // one filter row down
    auto mfma = [&](const bf16x8& a, const bf16x8& b, const floatx16& c) {
        if (TR) return mfma_x3<F16>(b, a, c);
        return mfma_x3<F16>(a, b, c);
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
    const unsigned out_np16 = OUT_HL ? p.out_np * 16u : 0u;
    const unsigned out_bytes = OUT_HL && TR ? p.out_np * (unsigned)ep.cout * 4u
                                            : (TR ? (unsigned)M * (unsigned)ep.cout * 4u : (unsigned)(M >> 1) * (unsigned)ep.cout * 4u);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, (int)out_bytes, 0x00020000);
    constexpr unsigned E_INVALID = 0xFFFF0000u;      // (the host keeps the output below 0xFFF00000 bytes)
    int rowb = ep.cout * 4;                          // bytes per output row
    float ebias[4] = {0.f, 0.f, 0.f, 0.f};
    if (!TR) {
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) ebias[cb] = ep.bias[n0 + 32 * cb + li < ep.cout ? n0 + 32 * cb + li : 0];
    }
    if (TR && tid < 128) *(LdsW4)(sBias + (unsigned)(tid * 4)) = __float_as_uint(ep.bias[n0 + tid < ep.cout ? n0 + tid : 0]);     // (visible after the prologue's barrier)
    if (tid < 64) *(LdsW4)(sProg + (unsigned)(tid * 4)) = 0u;
    const int wrow = TR ? wv * 64 + li : wv * 64 + 4 * lh;     // the lane's first row inside the tile (row block 0, group 0)
    float4 e_bb[2], e_v = make_float4(0.f, 0.f, 0.f, 0.f);
    e_bb[0] = e_v; e_bb[1] = e_v;
    float e_p0 = 0.f, e_p1 = 0.f;
    typedef const f32x4 __attribute__((address_space(3)))* LdsRF4;
    const unsigned bias_rd = sBias + (unsigned)(lh * 16);
    auto epi0_a = [&](int unit) {                    // the bias of unit `unit` (channels 32 cb + 8 g + 4 lh + {0..3}) into set unit & 1
        const int cb = (unit >> 2) & 3, g = unit & 3;
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
    // OUT_HL: x = hi + lo per value, as the consumers split an f32 input (conv_common.h split4).  Written out: hipcc converts the
    // hi parts twice and SLP-packs the residuals into v_pk_add_f32 (an anti-lever beside MFMAs, MI355X_MICROARCH.md): 12 VALU per unit
    unsigned e_h01 = 0, e_h23 = 0, e_l01 = 0, e_l23 = 0;
    auto cvt_pk = [&](float a, float b) { return cvt_pk16<F16>(a, b); };
    auto epi0_h = [&]() {                            // hi parts; residuals of the first pair
        e_h01 = cvt_pk(e_v.x, e_v.y); e_h23 = cvt_pk(e_v.z, e_v.w);
        e_v.x = e_v.x - unpk16_lo<F16>(e_h01);
        asm volatile("" : "+v"(e_v.x));
        e_v.y = e_v.y - unpk16_hi<F16>(e_h01);
        asm volatile("" : "+v"(e_v.y));
    };
    auto epi0_l = [&]() {                            // residuals of the second pair; lo parts
        e_v.z = e_v.z - unpk16_lo<F16>(e_h23);
        asm volatile("" : "+v"(e_v.z));
        e_v.w = e_v.w - unpk16_hi<F16>(e_h23);
        asm volatile("" : "+v"(e_v.w));
        e_l01 = cvt_pk(e_v.x, e_v.y); e_l23 = cvt_pk(e_v.z, e_v.w);
    };
    int e_lim = 0;                                   // rows of the tile that exist - the lane's first row: row block rb is stored iff rb * 32 < e_lim
    unsigned e_inv = E_INVALID;
    // kind 0, f32 output.  vb: byte offset of (row tile * tmr + wv * 64 + li, channel n0 + 4 lh) in `out`
    // OUT_HL: vb = CHL byte offset of (pixel tile * tmr + wv * 64 + li, chunk n0 / 16, k-half 0, hi) + 8 lh
    auto epi0_c = [&](int rb, int cb, int g, unsigned vb, int tile_rows) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if (OUT_HL) {
            unsigned sel;                            // (rb * 32 < e_lim) ? vb : E_INVALID (the invalid offset waits in a register: a literal beside vcc would be a
            asm volatile("v_cmp_lt_i32 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %1, vcc" : "=v"(sel) : "v"(vb), "n"(rb * 32), "v"(e_lim), "v"(e_inv) : "vcc");     // second constant-bus operand)
            if (X_NOEPI) return;
            // channels n0 + 32 cb + 8 g + 4 lh + {0..3}: chunk 2 cb + (g >> 1), k-half g & 1 -> plane 8 cb + 4 (g >> 1) + 2 (g & 1) (+ 1: lo)
            unsigned np16o = out_np16;
            asm volatile("" : "+s"(np16o));
            const unsigned so = (unsigned)(8 * cb + 4 * (g >> 1) + 2 * (g & 1)) * np16o;
            const unsigned vo = sel + (unsigned)(rb * 32 * 16);     // (immediate field; E_INVALID + 512 is still beyond the tensor)
            u32x2 dh, dl;
            dh[0] = e_h01; dh[1] = e_h23; dl[0] = e_l01; dl[1] = e_l23;
            __builtin_amdgcn_raw_buffer_store_b64(dh, orsrc, (int)vo, (int)so, 0);
            __builtin_amdgcn_raw_buffer_store_b64(dl, orsrc, (int)vo, (int)(so + np16o), 0);
        } else {
            int wr = wrow;
            asm volatile("" : "+v"(wr), "+s"(rowb));
            const bool ok = wr < tile_rows - rb * 32;
            const unsigned off = ok ? vb : E_INVALID;
            if (X_NOEPI) return;
            u32x4 d;
            d[0] = __float_as_uint(e_v.x); d[1] = __float_as_uint(e_v.y); d[2] = __float_as_uint(e_v.z); d[3] = __float_as_uint(e_v.w);
            __builtin_amdgcn_raw_buffer_store_b128(d, orsrc, (int)off, rb * 32 * rowb + (32 * cb + 8 * g) * 4, 0);
        }
    };
    // kind 1, unit (rb, cb, g): rows 8 g + 4 lh + {0..3} of the row block = two 2 x 1 pool windows, column n0 + 32 cb + li
    auto epi1_a = [&](const floatx16& acc, int cb, int g) {
        e_p0 = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]) + ebias[cb], 0.f);
        e_p1 = fmaxf(fmaxf(acc[4 * g + 2], acc[4 * g + 3]) + ebias[cb], 0.f);
        asm volatile("" : "+v"(e_p0), "+v"(e_p1));
    };
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
    // ---- kind 1, OUT_HL: the pooled output in the pixel-major split layout conv_dhl_kernel reads ("PHL"): pooled pixel P (NHWC order, so
    // that a window's pixels x channels ARE its flattened features) = Cout / 8 groups of [hi 8 x 16 bit | lo 8 x 16 bit] = the same
    // 4 bytes per element as f32.  Channel c of pixel P: byte P * Cout * 4 + (c >> 3) * 32 + (c & 7) * 2, lo part 16 bytes further.  The 32
    // lanes of a unit (32 channels of one pixel) fill one 128-byte line with their hi and lo stores.
    unsigned d_h0 = 0, d_l0 = 0, d_h1 = 0, d_l1 = 0;
    auto epi1_split = [&]() {
        d_h0 = cvt_pk16<F16>(e_p0, e_p0); d_h1 = cvt_pk16<F16>(e_p1, e_p1);
        const float r0 = e_p0 - unpk16_lo<F16>(d_h0), r1 = e_p1 - unpk16_lo<F16>(d_h1);
        d_l0 = cvt_pk16<F16>(r0, r0); d_l1 = cvt_pk16<F16>(r1, r1);
    };
    // vb: byte offset of (pooled row (tile * tmr + wv * 64) / 2 + 2 lh, channel n0 + li) -- hi part
    auto epi1_c = [&](int rb, int cb, int g, unsigned vb, int tile_rows) {
        int wr = wrow;
        asm volatile("" : "+v"(wr), "+s"(rowb));
        const bool ok = wr < tile_rows - (rb * 32 + 8 * g);
        const unsigned off = ok ? vb : E_INVALID;
        if (X_NOEPI) return;
        // unit (rb, g): pooled rows + rb * 16 + 4 g (+ 1); channel 32 cb + li: 4 cb groups of 32 bytes further on
        __builtin_amdgcn_raw_buffer_store_b16((short)d_h0, orsrc, (int)off, (rb * 16 + 4 * g) * rowb + cb * 128, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)d_l0, orsrc, (int)off, (rb * 16 + 4 * g) * rowb + cb * 128 + 16, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)d_h1, orsrc, (int)off, (rb * 16 + 4 * g + 1) * rowb + cb * 128, 0);
        __builtin_amdgcn_raw_buffer_store_b16((short)d_l1, orsrc, (int)off, (rb * 16 + 4 * g + 1) * rowb + cb * 128 + 16, 0);
    };
    auto epi_base = [&](int tile) {
        if (OUT_HL && !TR) return (unsigned)((tile * (TMR >> 1) + wv * 32 + 2 * lh) * ep.cout * 4 + ((n0 + li) >> 3) * 32 + ((n0 + li) & 7) * 2);
        if (OUT_HL) return ((unsigned)((n0 >> 4) * 4) * p.out_np + (unsigned)(tile * TMR + wv * 64 + li)) * 16u + (unsigned)(8 * lh);
        if (TR) return (unsigned)(((tile * TMR + wv * 64 + li) * ep.cout + n0 + 4 * lh) * 4);
        return (unsigned)(((tile * (TMR >> 1) + wv * 32 + 2 * lh) * ep.cout + n0 + li) * 4);
    };
    auto tile_rows_of = [&](int tile) { const int r = M - tile * TMR; return tile < ntiles ? (r < TMR ? r : TMR) : 0; };

    // ---- progress words: wave w has read tap v (0..6) of the second block of pass n for the last time -> word (v, w) = n
    auto publish = [&](int v, unsigned n) {          // v: compile-time
        if (lane == 0) *(LdsW4)(sProg + (unsigned)(v * 16) + (unsigned)(wv * 4)) = n;
    };
    typedef unsigned u32x4l __attribute__((ext_vector_type(4)));
    typedef const u32x4l __attribute__((address_space(3)))* LdsRU4;
    auto await = [&](int v, unsigned n) {            // every wave has published n for tap v (normally true at the first look)
        for (;;) {
            const u32x4l w = *(LdsRU4)(sProg + (unsigned)(v * 16));
            const unsigned m01 = w[0] < w[1] ? w[0] : w[1], m23 = w[2] < w[3] ? w[2] : w[3];
            const unsigned mn = m01 < m23 ? m01 : m23;
            if (__builtin_amdgcn_readfirstlane((int)mn) >= (int)n) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // ---- prologue
    int plo[G + 2];                                  // first pixels: the group's tiles + the next group's first two tiles
    unsigned lb[G][2];
    unsigned lbn[2];
    auto group_geometry = [&](int g0) {
        const GeoArgs ga = geo_args();
#pragma unroll
        for (int t = 0; t < G; ++t) {
            plo[t] = geo_plo(ga, g0 * G + t);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) lb[t][rb] = geo_lane(ga, g0 * G + t, rb, plo[t], t);
        }
    };
    group_geometry(grp);
    plo[G] = plo[0]; plo[G + 1] = plo[1];
    lbn[0] = lb[0][0]; lbn[1] = lb[0][1];
#pragma unroll
    for (int q = 0; q < WQ3_NFV; ++q) load_slice(plo[0], 0, q, 0);
#pragma unroll
    for (int v = 0; v < NT; ++v) { load_weight_piece(0, v, 0, 0u); load_weight_piece(0, v, 1, 0u); }
    wait_vmcnt<0>();
    __syncthreads();

    AFr a[2][2];
    bf16x8 bh[2][4], bl[4], blast[4];
    unsigned ra[2];                                  // running addresses (filter row of the tap being read)
    auto read_a_h = [&](AFr& f, unsigned ad, int kx) { f.h = *(LdsR16)(ad + (unsigned)(kx * 16)); };
    auto read_a_l = [&](AFr& f, unsigned ad, int kx) { f.l = *(LdsR16)(ad + (unsigned)(kx * 16 + WQH_PL)); };
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        ra[rb] = lb[0][rb];
        read_a_h(a[0][rb], ra[rb], 0); read_a_l(a[0][rb], ra[rb], 0);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) bh[0][cb] = *(LdsR16)(b_addr(0, cb));

    const int nchunk = p.Cin / F2_CH;                // >= 2 (host-checked)
    const int gstep = (int)gridDim.x;
    int prev_tile1 = ntiles;
    unsigned pass = 0;                               // chunk passes of this workgroup so far (the weights resident now belong to pass `pass`)
    for (; grp < ngroups; grp += gstep) {
        const bool last_group = grp + gstep >= ngroups;
        auto run_block = [&](const int t, const bool ZC, const bool EP, const int c0, const bool last_chunk,
                             floatx16& d00, floatx16& d01, floatx16& d02, floatx16& d03,
                             floatx16& d10, floatx16& d11, floatx16& d12, floatx16& d13,
                             const floatx16& o00, const floatx16& o01, const floatx16& o02, const floatx16& o03,
                             const floatx16& o10, const floatx16& o11, const floatx16& o12, const floatx16& o13,
                             const int etile) __attribute__((always_inline)) {
            const int pn = t == 0 ? plo[1] : (last_chunk ? plo[G] : plo[0]);        // the tile whose footprint this block fetches
            const int nc0 = t == 0 ? c0 : (last_chunk ? 0 : c0 + F2_CH);          // ... and its chunk = the chunk of the weights fetched (t == 1)
            const unsigned sw_next = (pass & 1u) ? 0u : (unsigned)(2 * WQH_SLOT);    // taps 7, 8 of pass + 1
            unsigned vb = E_INVALID;
            int erows = 0;
            if (EP) { vb = epi_base(etile); erows = tile_rows_of(etile); if (OUT_HL) { e_lim = erows - wrow; e_inv = E_INVALID; asm volatile("" : "+v"(e_inv)); }
                    }
#pragma unroll
            for (int v = 0; v < NT; ++v) {
                const int cs = (v + t) & 1, ns = cs ^ 1;
                const bool last = v + 1 == NT;
                const int ky1 = (v + 1) / KW, kx1 = (v + 1) % KW;
                if (last) {
                    // every fragment read of this block has been issued; this wave's DMA pieces (the other footprint, and in a
                    // chunk's second block the next chunk's weights) have landed; then meet
                    __builtin_amdgcn_sched_barrier(0);
                    if (!X_NOBAR) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (t == 1) {                    // the next chunk's weights are resident from here on
                        bread78 = bread + sw_next;
                        asm volatile("" : "+v"(bread78));
                    }
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
                    // 4..7: the A fragments of the next tap (a.l x 2, a.h x 2); 8..11: its hi weights (in the last step: the next
                    // block's first tap -- in a chunk's second block that is the NEXT chunk's, resident behind the barrier above)
                    if (s < 12) {
                        const int tn = last ? 0 : v + 1;
                        if (!last) {
                            if (s == 4 && kx1 == 0) {
#pragma unroll
                                for (int r = 0; r < 2; ++r) { asm volatile("" : "+v"(ra[r])); ra[r] += wstep; }
                            }
                        } else if (s == 4) {
#pragma unroll
                            for (int r = 0; r < 2; ++r) ra[r] = t == 0 ? lb[1][r] : (last_chunk ? lbn[r] : lb[0][r]);
                        }
                        const int kx = last ? 0 : kx1;
                        if (s < 4) { if (!last) bl[s] = *(LdsR16)(b_addr(v, s) + 2048); }
                        else if (s < 6) read_a_l(a[ns][s - 4], ra[s - 4], kx);
                        else if (s < 8) read_a_h(a[ns][s - 6], ra[s - 6], kx);
                        else bh[ns][s - 8] = *(LdsR16)(b_addr(tn, s - 8));
                    }
                    // the last tap's lo weights, in the step before it (slots 12..15)
                    if (v + 2 == NT && s >= 12 && s < 16) blast[s - 12] = *(LdsR16)(b_addr(NT - 1, s - 12) + 2048);
                    // ---- weights of the next chunk (a chunk's second block only).  Step 0: taps 7, 8 into the free slot pair; step v + 1
                    // (v = 0..6): tap v, which every wave has read for the last time in slot 3 of its step v (published in slot 5)
                    if (t == 1) {
                        if (v <= 6 && s == 5 && !X_NOFLAG) publish(v, pass + 1u);
                        if (v == 0 && s >= 12 && s < 16) load_weight_piece(nc0, 7 + ((s - 12) >> 1), (s - 12) & 1, sw_next);
                        if (v >= 1 && v <= 7 && s == 16 && !X_NOFLAG) await(v - 1, pass + 1u);
                        if (v >= 1 && v <= 7 && (s == 17 || s == 18)) load_weight_piece(nc0, v - 1, s - 17, 0u);
                    }
                    // ---- footprint of the next block: slices 2 v, 2 v + 1 in step v (v = 0..3), slots 20 and 22
                    if (v <= 3 && (s == 20 || s == 22)) load_slice(pn, nc0, 2 * v + ((s - 20) >> 1), 1 - t);
                    // ---- epilogue of the other accumulator set: 32 units (rb, cb, g)
                    // (the block's barrier waits vmcnt(0) for its LDS-DMA pieces, which also waits for every store issued before it:
                    // the units are packed into the first steps so that the last store is two to four steps old by then)
                    if (EP && TR && !OUT_HL && v <= 5 && s < 18) {            // three pieces per unit, six units per step
                        const int unit = v * 6 + s / 3;
                        if (unit < 32) {
                            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                            if (s % 3 == 0) { if (unit == 0) epi0_a(0); if (unit + 1 < 32) epi0_a(unit + 1); }
                            else if (s % 3 == 1) epi0_b(oa, unit);
                            else epi0_c(erb, ecb, eg, vb, erows);
                        }
                    }
                    if (EP && TR && OUT_HL && v <= 5) {                      // three pieces per unit in four slots, six units per step
                        const int unit = v * 6 + s / 4;
                        if (unit < 32) {
                            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                            if (s % 4 == 0) { if (unit == 0) epi0_a(0); if (unit + 1 < 32) epi0_a(unit + 1); epi0_b(oa, unit); }
                            else if (s % 4 == 1) epi0_h();
                            else if (s % 4 == 2) epi0_l();
                            else epi0_c(erb, ecb, eg, vb, erows);
                        }
                    }
                    if (EP && !TR && OUT_HL && v <= 5) {                     // three pieces per unit in four slots, six units per step
                        const int unit = v * 6 + s / 4;
                        if (unit < 32) {
                            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                            const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                            if (s % 4 == 0) epi1_a(oa, ecb, eg);
                            else if (s % 4 == 1) epi1_split();
                            else if (s % 4 == 2) epi1_c(erb, ecb, eg, vb, erows);
                        }
                    }
                    if (EP && !TR && !OUT_HL && v <= 3 && s < 16) {          // two pieces per unit, eight units per step
                        const int unit = v * 8 + s / 2;
                        const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
                        const floatx16& oa = erb == 0 ? (ecb == 0 ? o00 : ecb == 1 ? o01 : ecb == 2 ? o02 : o03) : (ecb == 0 ? o10 : ecb == 1 ? o11 : ecb == 2 ? o12 : o13);
                        if (s % 2 == 0) epi1_a(oa, ecb, eg); else epi1_b(erb, ecb, eg, vb, erows);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t == 1) ++pass;
        };
#define ISS_WQH_SET0 c000, c001, c002, c003, c010, c011, c012, c013
#define ISS_WQH_SET1 c100, c101, c102, c103, c110, c111, c112, c113
        run_block(0, true, true, 0, false, ISS_WQH_SET0, ISS_WQH_SET1, prev_tile1);
        run_block(1, true, false, 0, false, ISS_WQH_SET1, ISS_WQH_SET0, 0);
        for (int ch = 1; ch + 1 < nchunk; ++ch) {
            run_block(0, false, false, ch * F2_CH, false, ISS_WQH_SET0, ISS_WQH_SET1, 0);
            run_block(1, false, false, ch * F2_CH, false, ISS_WQH_SET1, ISS_WQH_SET0, 0);
        }
        {
            if (!last_group) {
                const GeoArgs ga = geo_args();
                plo[G] = geo_plo(ga, (grp + gstep) * G);
                plo[G + 1] = geo_plo(ga, (grp + gstep) * G + 1);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) lbn[rb] = geo_lane(ga, (grp + gstep) * G, rb, plo[G], 0);
            } else {
                plo[G] = plo[0]; plo[G + 1] = plo[1];
                lbn[0] = lb[0][0]; lbn[1] = lb[0][1];
            }
            run_block(0, false, false, (nchunk - 1) * F2_CH, true, ISS_WQH_SET0, ISS_WQH_SET1, 0);
            run_block(1, false, true, (nchunk - 1) * F2_CH, true, ISS_WQH_SET1, ISS_WQH_SET0, grp * G);
        }
        prev_tile1 = grp * G + 1;
        if (!last_group) {
            lb[0][0] = lbn[0]; lb[0][1] = lbn[1];
            plo[0] = plo[G]; plo[1] = plo[G + 1];
            const GeoArgs ga = geo_args();
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) lb[1][rb] = geo_lane(ga, (grp + gstep) * G + 1, rb, plo[1], 1);
        }
    }
    // ---- the last group's tile 1: the only serial epilogue of the workgroup
    {
        const unsigned vb = epi_base(prev_tile1);
        const int erows = tile_rows_of(prev_tile1);
        e_lim = erows - wrow; e_inv = E_INVALID; asm volatile("" : "+v"(e_inv));
#pragma unroll
        for (int unit = 0; unit < 32; ++unit) {
            const int erb = unit >> 4, ecb = (unit >> 2) & 3, eg = unit & 3;
            const floatx16& oa = erb == 0 ? (ecb == 0 ? c100 : ecb == 1 ? c101 : ecb == 2 ? c102 : c103) : (ecb == 0 ? c110 : ecb == 1 ? c111 : ecb == 2 ? c112 : c113);
            if (TR) { epi0_a(unit); epi0_b(oa, unit); if (OUT_HL) { epi0_h(); epi0_l(); } epi0_c(erb, ecb, eg, vb, erows); }
            else if (OUT_HL) { epi1_a(oa, ecb, eg); epi1_split(); epi1_c(erb, ecb, eg, vb, erows); }
            else { epi1_a(oa, ecb, eg); epi1_b(erb, ecb, eg, vb, erows); }
        }
    }
    wait_vmcnt<0>();                                 // (LDS-DMA pieces of a block that never ran must not outlive the workgroup's LDS)
#undef ISS_WQH_SET0
#undef ISS_WQH_SET1
}

// kind 0 = bias + relu (f32 NHWC or, a.out_hl, CHL output), kind 1 = relu + 2 x 1 max-pool (f32 output or, a.out_hl, the CHL input of a
// dense layer: conv_dhl.h); a.in_hl is implied
void iss_wq3h_launch(const ConvArgs& a, dim3 grid, hipStream_t st, int kind);

}  // namespace issk
