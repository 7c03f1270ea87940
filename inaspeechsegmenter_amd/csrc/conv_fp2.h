// conv_x3_fp2_kernel: second-generation LDS-footprint bf16x3 implicit GEMM (kh*kw*NH >= 6 virtual taps, Cin % 32 == 0).
//
// Same decomposition as conv_x3_fp_kernel (conv_fp.h): the input pixels a 128-row M tile touches are one contiguous range
// of the flattened (sample, iy, ix) index -- its "footprint" -- which is split into bf16 hi/lo once and kept in LDS while
// every filter tap reads its MFMA A fragments from it.  What changed, and why (profiles/r01_pmc.md: the first kernel
// issued 5.0 ordinary VALU instructions per MFMA, 3.5 of them in a serial convert-the-footprint phase between two
// barriers during which the workgroup issues no MFMA at all):
//
//   * the k loop walks 16-channel chunks, so a footprint is 352 pixels x (16 hi + 16 lo) bf16 = 28 KB and TWO of them fit
//     beside the weight ring at two workgroups per CU: the next chunk's footprint is converted slice by slice BEHIND the
//     MFMAs of the current chunk into the other buffer -- there is no serial staging phase and no extra barrier;
//   * footprint rows are 80 bytes (64 + 16 pad) and LINEAR: the A-fragment address of a tap is (lane base + tap offset),
//     one v_add per step instead of a swizzle computation (~12 VALU per tap), and 16 consecutive pixels still hit 16
//     different 16-byte bank groups (20 p mod 64 is a permutation of the multiples of 4);
//   * zero-padded taps read an all-zero pixel kept behind each footprint (one v_cndmask on the address instead of zeroing
//     eight fragment registers);
//   * a step = one tap x one 16-channel chunk = 6 MFMAs; a phase = 2 steps = 12 MFMAs between barriers (as before).
//     Weight tiles (64 rows x 16 k, hi + lo = 4 KB per step) arrive by LDS-DMA two phases ahead into a ring of six stages;
//     their 16-byte slots are permuted on the SOURCE side so that the ds_read_b128 of the 16 lanes of a group are
//     conflict-free (slot = 2 n + (h ^ ((n >> 3) & 1)));
//   * epilogue parameters are read through an opaque copy of the kernel-argument pointer at the end of a tile instead of
//     living in SGPRs through the main loop (the first kernel spilled 136 SGPRs and 11 VGPRs).
#pragma once
#include "conv_fp.h"

namespace issk {

constexpr int F2_PIX = 352;                       // footprint capacity in pixels (host-validated per launch)
constexpr int F2_ROW = 80;                        // bytes per footprint pixel: 16 ch hi (32 B) | 16 ch lo (32 B) | 16 B pad
constexpr int F2_ZERO = F2_PIX * F2_ROW;          // byte offset of the all-zero pixel behind a footprint
constexpr int F2_BUF = (F2_PIX + 1) * F2_ROW;     // bytes of one footprint buffer (28 240)
constexpr int F2_NFV = (F2_PIX + 63) / 64;        // 64-pixel slices per footprint (6)
constexpr int F2_BST = 4096;                      // bytes of one weight stage: hi plane (64 rows x 16 k bf16 = 2 KB) | lo plane
constexpr int F2_NST = 6;                         // weight stages: three phases of two steps
constexpr int F2_CH = 16;                         // channels per chunk (one k16 MFMA step)
constexpr int F2_LDS = 2 * F2_BUF + F2_NST * F2_BST;    // 81 056 bytes: two workgroups per CU

typedef const bf16x8 __attribute__((address_space(3)))* LdsR16;
typedef bf16x4 __attribute__((address_space(3)))* LdsW8;
typedef unsigned __attribute__((address_space(3)))* LdsW4;

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
};

template <int KH, int KW, bool PADDED, bool TR, bool FUSED = false, int NH = 1>
__global__ __launch_bounds__(256, 2) void conv_x3_fp2_kernel(const ConvArgs p) {
    constexpr int NT = KH * KW * NH;                 // virtual taps (steps) per 16-channel chunk
    constexpr int NSTEP = 2 * NT;                    // steps per unrolled iteration: two chunks, footprint buffers 0 and 1
    constexpr int NPH = NT;                          // phases (2 steps) per iteration
    static_assert(NH == 1 || NH == 2, "");
    static_assert(NT >= 8, "too few taps to hide the footprint conversion; use conv_x3_fp_kernel");
    static_assert(NH == 1 || !FUSED, "");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[F2_LDS];    // [footprint 0][footprint 1][6 weight stages]
    const unsigned sF_base = (unsigned)(size_t)smem;                         // LDS byte addresses (the low half of the flat address)
    const unsigned sB_base = sF_base + 2 * F2_BUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * BN * NH;
    const int li = lane & 31, lh = lane >> 5;
    const int M = (int)p.M;
    int totpix;                                      // samples * H * W
    { const int spp = p.Hq * p.Wq * p.pp; totpix = (int)(p.img_stride / p.Cin) * (M / spp); }

    const int per = ((int)p.nblk + (int)gridDim.x - 1) / (int)gridDim.x;
    constexpr bool inter = FUSED;                    // shared first layer: interleaved tile order (see conv_fp.h)
    const int tstep = inter ? (int)gridDim.x : 1;
    int tile = inter ? (int)blockIdx.x : (int)blockIdx.x * per;
    const int tile_end = inter ? (int)p.nblk : (tile + per < (int)p.nblk ? tile + per : (int)p.nblk);
    if (tile >= tile_end) return;

    // ---- per-tile geometry (as conv_x3_fp_kernel): first pixel of the tile, pixels it needs, this lane's A row
    struct Geom { int p_lo, need, lanepix; unsigned vmask; int fy, fx; bool two; };
    struct Win { int wr0, wr1; float mean0, mean1, sd0, sd1; int live0, live1; };
    // the row-decomposition parameters are read through an opaque copy of the kernel-argument pointer where a tile's
    // geometry is computed (once per tile), so that they do not occupy SGPRs through the main loop
    auto geo_args = [&]() {
        KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();          // the by-value ConvArgs is the kernel's only argument
        asm volatile("" : "+s"(q));
        GeoArgs ga;
        ga.H = q->H; ga.W = q->W; ga.Hq = q->Hq; ga.Wq = q->Wq; ga.ph = q->ph; ga.pw = q->pw; ga.pp = q->pp;
        ga.sh = q->sh; ga.sw = q->sw; ga.pt_ = q->pt_; ga.pl_ = q->pl_;
#pragma unroll
        for (int i = 0; i < 4; ++i) { ga.dv_mul[i] = q->dv_mul[i]; ga.dv_sh[i] = q->dv_sh[i]; }
        return ga;
    };
    auto geometry = [&](int t) {
        const GeoArgs ga = geo_args();
        Geom g;
        const int m0 = t * BM;
        int b, oy, ox;
        map_row32(ga, m0, b, oy, ox);
        g.p_lo = (b * ga.H + (oy * ga.sh - ga.pt_)) * ga.W + (ox * ga.sw - ga.pl_);
        g.fy = 0; g.fx = 0; g.two = false;
        if (FUSED) { g.fy = oy * ga.sh; g.fx = ox * ga.sw; }
        {
            const int ml = m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1;
            int b2, oy2, ox2;
            map_row32(ga, ml, b2, oy2, ox2);
            g.need = (b2 * ga.H + (oy2 * ga.sh - ga.pt_ + KH - 1)) * ga.W + (ox2 * ga.sw - ga.pl_ + KW - 1) - g.p_lo + 1;
        }
        if (FUSED) g.two = g.fy * ga.W + g.fx + g.need > ga.H * ga.W;
        const int m = m0 + wv * 32 + li;
        map_row32(ga, m < M ? m : m0, b, oy, ox);
        const int iy0 = oy * ga.sh - ga.pt_, ix0 = ox * ga.sw - ga.pl_;
        const int lanepix = (b * ga.H + iy0) * ga.W + ix0 - g.p_lo;
        const int hi = F2_PIX - 1 - ((KH - 1) * ga.W + (KW - 1));    // keeps every tap of a row >= M inside the buffer
        g.lanepix = lanepix < 0 ? 0 : (lanepix > hi ? hi : lanepix);
        g.vmask = 0xffffffffu;
        if (PADDED) {                                // bit (ky * KW + kx): the tap reads inside the image
            unsigned vm = 0;
#pragma unroll
            for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                for (int kx = 0; kx < KW; ++kx)
                    vm |= ((unsigned)(iy0 + ky) < (unsigned)ga.H && (unsigned)(ix0 + kx) < (unsigned)ga.W) ? 1u << (ky * KW + kx) : 0u;
            g.vmask = vm;
        }
        return g;
    };
    auto windows_of = [&](int t) {                   // loads only: nothing here may USE the values (see conv_fp.h)
        const GeoArgs ga = geo_args();
        Win w;
        int b, oy, ox;
        map_row32(ga, t * BM, b, oy, ox);
        const int nb = M / (ga.Hq * ga.Wq * ga.pp);
        const unsigned b0 = (unsigned)(b < nb ? b : nb - 1), b1 = (unsigned)(b + 1 < nb ? b + 1 : nb - 1);
        w.wr0 = p.win_row[b0]; w.mean0 = p.stats[2u * b0]; w.sd0 = p.stats[2u * b0 + 1u]; w.live0 = p.finite[b0];
        w.wr1 = p.win_row[b1]; w.mean1 = p.stats[2u * b1]; w.sd1 = p.stats[2u * b1 + 1u]; w.live1 = p.finite[b1];
        return w;
    };
    auto settle = [&](Win& w) {
        asm volatile("" : "+v"(w.wr0), "+v"(w.wr1), "+v"(w.mean0), "+v"(w.mean1), "+v"(w.sd0), "+v"(w.sd1), "+v"(w.live0), "+v"(w.live1));
    };
    Geom g = geometry(tile), gn = g;
    Win wc = {}, wn = {}, wx = {};                   // current tile (settled), next tile (pending), target of fetch / convert
    if (FUSED) { wc = windows_of(tile); settle(wc); wx = wc; }

    // ---- weight tiles by LDS-DMA: one 1 KB piece per wave and step.  Wave w fills plane w >> 1 (hi / lo), half w & 1;
    // lane l writes 16-byte slot s = 64 (w & 1) + l of that plane and fetches the (row n, k half h) that belongs there:
    // n = s >> 1, h = (s & 1) ^ ((n >> 3) & 1).  Rows >= Cout read row 0 (their output columns are never stored).
    const uint16_t* const wsrc = (wv >> 1) ? p.wl : p.wh;
    unsigned boff;
    {
        const int s = 64 * (wv & 1) + lane, n = s >> 1, h = (s & 1) ^ ((n >> 3) & 1);
        boff = 2u * ((unsigned)(n0 + n < p.Cout ? n0 + n : 0) * (unsigned)p.Kpad + (unsigned)(h * 8));       // bytes
    }
    auto dma_b = [&](int stage_off, int vtap, int c0) {          // vtap: compile-time virtual tap, c0: first channel of the chunk
        const unsigned k = (unsigned)((vtap / NH) * p.Cin + c0) + (NH > 1 ? (unsigned)((vtap % NH) * BN * p.Kpad) : 0u);
        glds16(reinterpret_cast<const char*>(wsrc) + 2u * k, boff, (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_base + stage_off + wv * 1024)));
    };
    // B fragment of this lane: row n = li (and li + 32: + 1024 bytes), k half lh
    const unsigned bread = sB_base + (unsigned)((2 * li + (lh ^ ((li >> 3) & 1))) * 16);

    floatx16 acc0, acc1, acc2, acc3;                 // acc2 / acc3: second 64-column half (NH = 2 only)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; acc2[i] = 0.f; acc3[i] = 0.f; }

    // ---- footprint slices: thread -> pixel 64 q + (tid >> 2), channels [c0 + 4 (tid & 3), + 4)
    const int cg = tid & 3, prow = tid >> 2;
    float4 fv[F2_NFV];
    unsigned dbmask = 0;                             // FUSED: bit q = slice pixel belongs to the second window
    float4 fsw = make_float4(0.f, 0.f, 0.f, 0.f), fbw = fsw, fps = fsw, fpt = fsw;
    const int magicW = (65536 + p.W - 1) / p.W;      // x / W == (x * magicW) >> 16 for x < 512, W <= 128
    auto fetch_slice = [&](int q, const Geom& gg, int c0) {
        const int qq = 64 * q < gg.need ? q : 0;     // unneeded slices re-load slice 0 (the load COUNT must not change)
        if (FUSED) {
            int x = gg.fx + prow + 64 * qq;
            const int dy = (x * magicW) >> 16;
            x -= dy * p.W;
            int y = gg.fy + dy;
            const bool second = y >= p.H;
            y -= second ? p.H : 0;
            const int row = y + (second ? wx.wr1 : wx.wr0) - p.f_rmin;
            dbmask = (q == 0 ? 0u : dbmask) | (second ? 1u << q : 0u);
            fv[q] = *reinterpret_cast<const float4*>(p.in + ((unsigned)(row * p.W + x) * (unsigned)p.Cin + (unsigned)(c0 + cg * 4)));
            return;
        }
        int gp = gg.p_lo + prow + 64 * qq;
        gp = gp < 0 ? 0 : (gp > totpix - 1 ? totpix - 1 : gp);
        fv[q] = *reinterpret_cast<const float4*>(p.in + ((unsigned)gp * (unsigned)p.Cin + (unsigned)(c0 + cg * 4)));
    };
    auto fetch_chan = [&](int c0) {                  // FUSED: first-layer constants of this thread's 4 channels
        const unsigned o = (unsigned)(c0 + cg * 4);
        fsw = *reinterpret_cast<const float4*>(p.f_wsum + o);
        fbw = *reinterpret_cast<const float4*>(p.f_bias + o);
        fps = *reinterpret_cast<const float4*>((p.f_ps ? p.f_ps : p.f_bias) + o);
        fpt = *reinterpret_cast<const float4*>((p.f_pt ? p.f_pt : p.f_bias) + o);
    };
    constexpr int NCHLD = 4;                         // loads of fetch_chan
    const float f_lob = p.f_act == 1 ? 0.f : -INFINITY;             // FUSED: lower bound of the first layer's activation (relu / none)
    const bool f_has_ps = p.f_ps != nullptr;
    // per-chunk constants of the conversion (FUSED): scale 1 / std and shift bias - mean / std * sum_k w per window
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f}, rs0 = 0.f, rs1 = 0.f;
    float4 cps = fsw, cpt = fsw;                     // post-activation scale / shift of the chunk being converted
    auto conv_consts = [&]() {
        if (!FUSED) return;
        asm volatile("" : "+v"(fsw.x), "+v"(fsw.y), "+v"(fsw.z), "+v"(fsw.w), "+v"(fbw.x), "+v"(fbw.y), "+v"(fbw.z), "+v"(fbw.w));
        asm volatile("" : "+v"(fps.x), "+v"(fps.y), "+v"(fps.z), "+v"(fps.w), "+v"(fpt.x), "+v"(fpt.y), "+v"(fpt.z), "+v"(fpt.w));
        rs0 = wx.live0 ? 1.0f / wx.sd0 : 0.f;
        rs1 = wx.live1 ? 1.0f / wx.sd1 : 0.f;
        const float mr0 = wx.live0 ? -wx.mean0 * rs0 : 0.f, mr1 = wx.live1 ? -wx.mean1 * rs1 : 0.f;
        const float sw[4] = {fsw.x, fsw.y, fsw.z, fsw.w}, bw[4] = {fbw.x, fbw.y, fbw.z, fbw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { t0[i] = fmaf(sw[i], mr0, bw[i]); t1[i] = fmaf(sw[i], mr1, bw[i]); }
        cps = f_has_ps ? fps : make_float4(1.f, 1.f, 1.f, 1.f);
        cpt = f_has_ps ? fpt : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // convert slice q (registers fv[q]) and write it into footprint buffer `buf` (compile-time)
    auto convert_slice = [&](int q, int buf) {
        // every slice is converted (slices the tile does not need hold slice 0's data, see fetch_slice): a uniform skip
        // would put a branch with loads in flight into the phase, and the work is hidden behind the MFMAs anyway.
        // Opaque pass-through: the conversion cannot be scheduled above this point (hipcc hoisted its FMAs right behind
        // the loads, i.e. an s_waitcnt for fresh loads in front of the MFMAs the loads are meant to hide behind).
        float4 v = fv[q];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(dbmask));
        if (FUSED) {
            const bool second = (dbmask >> q) & 1u;
            const float sc = second ? rs1 : rs0;
            v = make_float4(fmaf(v.x, sc, second ? t1[0] : t0[0]), fmaf(v.y, sc, second ? t1[1] : t0[1]),
                            fmaf(v.z, sc, second ? t1[2] : t0[2]), fmaf(v.w, sc, second ? t1[3] : t0[3]));
            v.x = fmaxf(v.x, f_lob); v.y = fmaxf(v.y, f_lob); v.z = fmaxf(v.z, f_lob); v.w = fmaxf(v.w, f_lob);
            // post-activation affine of the first layer (conv -> relu -> BatchNorm); scale 1 / shift 0 when absent
            v.x = v.x * cps.x + cpt.x; v.y = v.y * cps.y + cpt.y; v.z = v.z * cps.z + cpt.z; v.w = v.w * cps.w + cpt.w;
        }
        bf16x4 h, l;
        split4(v, h, l);
        const unsigned dst = sF_base + (unsigned)(buf * F2_BUF + (prow + 64 * q) * F2_ROW + cg * 8);
        if (64 * q + 63 < F2_PIX || prow + 64 * q < F2_PIX) {            // (only the last slice has pixels beyond the buffer)
            *(LdsW8)(dst) = h;
            *(LdsW8)(dst + 32) = l;
        }
    };

    // ---- fragments
    struct AFr { bf16x8 h, l; };
    struct BFr { bf16x8 b0h, b0l, b1h, b1l; };
    auto read_a = [&](AFr& f, const Geom& gg, int buf, int tap) {        // buf, tap: compile-time
        const int ky = tap / KW, kx = tap % KW;
        unsigned a = (unsigned)((gg.lanepix + ky * p.W + kx) * F2_ROW + lh * 16);
        if (PADDED) a = (gg.vmask >> tap) & 1u ? a : (unsigned)F2_ZERO;   // zero-padded tap: the all-zero pixel
        a += sF_base + (unsigned)(buf * F2_BUF);
        f.h = *(LdsR16)(a);
        f.l = *(LdsR16)(a + 32);
    };
    auto read_b = [&](BFr& f, int stage_off) {
        const unsigned a = bread + (unsigned)stage_off;
        f.b0h = *(LdsR16)(a);
        f.b1h = *(LdsR16)(a + 1024);
        f.b0l = *(LdsR16)(a + 2048);
        f.b1l = *(LdsR16)(a + 3072);
    };
#define ISS_F2_MFMA6(A0, A1)                                                                        \
    if (TR) {                                            /* C^T: rows = channels, columns = pixels (epilogue_tr) */ \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b0h, a.l, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b1h, a.l, A1, 0, 0, 0);                      \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b0l, a.h, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b1l, a.h, A1, 0, 0, 0);                      \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b0h, a.h, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.b1h, a.h, A1, 0, 0, 0);                      \
    } else {                                                                                        \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.b0h, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.b1h, A1, 0, 0, 0);                      \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.b0l, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.b1l, A1, 0, 0, 0);                      \
        A0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.b0h, A0, 0, 0, 0);                      \
        A1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.b1h, A1, 0, 0, 0);                      \
    }
    auto mfma6 = [&](const AFr& a, const BFr& b, int half) {             // half: compile-time
        if (NH == 1 || half == 0) { ISS_F2_MFMA6(acc0, acc1) } else { ISS_F2_MFMA6(acc2, acc3) }
    };
#undef ISS_F2_MFMA6

    // ---- schedule of one unrolled iteration (chunks cc = 0, 1; steps s = cc * NT + v; phase ph = steps 2 ph, 2 ph + 1):
    //   chunk cc reads footprint buffer cc.  During its steps the footprint of the NEXT chunk is built in buffer 1 - cc:
    //   all its slices are fetched (global -> registers) in the phase P0(cc) that contains the chunk's first step and
    //   converted k per phase in the phases after it, finishing one barrier before the first A read of the next chunk
    //   (the A fragments of step s are read during step s - 1).
    constexpr int P0_0 = 0, P0_1 = NT / 2;
    constexpr int PL_0 = (NT - 1) / 2 - 1, PL_1 = NT - 2;                // last phases in which a conversion may happen
    constexpr int CD = 2;                                                // phases between a slice's fetch and its conversion
    constexpr int NAV_0 = PL_0 - P0_0 - (CD - 1), NAV_1 = PL_1 - P0_1 - (CD - 1);
    static_assert(NAV_0 >= 1 && NAV_1 >= 1, "");
    constexpr int K_0 = (F2_NFV + NAV_0 - 1) / NAV_0, K_1 = (F2_NFV + NAV_1 - 1) / NAV_1;     // slices per phase

    // ---- prologue: zero pixels, first footprint (chunk 0 of the first tile) converted serially, first weight stages
    if (tid < 2 * (F2_ROW / 4)) *(LdsW4)(sF_base + (unsigned)((tid / (F2_ROW / 4)) * F2_BUF + F2_ZERO + (tid % (F2_ROW / 4)) * 4)) = 0u;
    if (FUSED) fetch_chan(0);
#pragma unroll
    for (int q = 0; q < F2_NFV; ++q) fetch_slice(q, g, 0);
    int st[F2_NST];
#pragma unroll
    for (int i = 0; i < F2_NST; ++i) st[i] = i * F2_BST;                 // st[s % 6]: stage of step s of the iteration
    dma_b(st[0], 0 % NT, (0 / NT) * F2_CH);
    dma_b(st[1], 1 % NT, (1 / NT) * F2_CH);
    dma_b(st[2], 2 % NT, (2 / NT) * F2_CH);
    dma_b(st[3], 3 % NT, (3 / NT) * F2_CH);
    conv_consts();
#pragma unroll
    for (int q = 0; q < F2_NFV; ++q) convert_slice(q, 0);
    wait_vmcnt<0>();
    __syncthreads();
    AFr a0, a1;
    BFr b0, b1;
    read_a(a0, g, 0, 0);

    const int nsc = p.Cin / (2 * F2_CH);             // unrolled iterations (two chunks each) per tile
    __builtin_amdgcn_s_setprio(2);
    for (; tile < tile_end; tile += tstep) {
        const bool last_tile = tile + tstep >= tile_end;
        for (int sc = 0; sc < nsc; ++sc) {
            const int c0 = sc * 2 * F2_CH;           // first channel of chunk 0 of this iteration
            const bool last_sc = sc + 1 == nsc;
            if (sc == (nsc >= 2 ? nsc - 2 : 0) && !last_tile) {          // next tile's geometry one iteration early
                gn = geometry(tile + tstep);
                if (FUSED) wn = windows_of(tile + tstep);
            }
            if (sc == (nsc >= 2 ? nsc - 2 : 0) && last_tile) gn = g;    // nothing follows: re-stage this tile (never read)
            // chunk 1 of the tile's last iteration builds chunk 0 of the NEXT tile
            const Geom gx1 = last_sc ? gn : g;
            const int nx1_c0 = last_sc ? 0 : c0 + 2 * F2_CH;
            Win wx1 = wc;                            // FUSED: windows of the footprint chunk 1 builds
            if (FUSED && last_sc && !last_tile) { settle(wn); wx1 = wn; }
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) {
                const int s0 = 2 * ph, s1 = 2 * ph + 1;
                const int cc0 = s0 / NT, cc1 = s1 / NT;                 // chunk (= footprint buffer) of each step
                const int v0 = s0 % NT, v1 = s1 % NT;
                // -- global loads of this phase (compile-time count)
                const bool f0 = ph == P0_0, f1 = ph == P0_1;
                const int nld = (f0 || f1) ? F2_NFV + (FUSED ? NCHLD : 0) : 0;
                if (f0) {
                    if (FUSED) { wx = wc; fetch_chan(c0 + F2_CH); }
#pragma unroll
                    for (int q = 0; q < F2_NFV; ++q) fetch_slice(q, g, c0 + F2_CH);
                }
                if (f1) {
                    if (FUSED) { wx = wx1; fetch_chan(nx1_c0); }
#pragma unroll
                    for (int q = 0; q < F2_NFV; ++q) fetch_slice(q, gx1, nx1_c0);
                }
                // -- weight tiles of phase ph + 2 (steps s0 + 4, s1 + 4; beyond this iteration: the next one's first steps)
                {
                    const int t0s = s0 + 4, t1s = s1 + 4;
                    const int ca = t0s < NSTEP ? c0 + (t0s / NT) * F2_CH : (last_sc ? 0 : c0 + 2 * F2_CH) + ((t0s - NSTEP) / NT) * F2_CH;
                    const int cb = t1s < NSTEP ? c0 + (t1s / NT) * F2_CH : (last_sc ? 0 : c0 + 2 * F2_CH) + ((t1s - NSTEP) / NT) * F2_CH;
                    dma_b(st[t0s % F2_NST], (t0s % NSTEP) % NT, ca);
                    dma_b(st[t1s % F2_NST], (t1s % NSTEP) % NT, cb);
                }
                // -- step s0: its A fragments were read during the previous step, its weights became visible at the barrier
                read_b(b0, st[s0 % F2_NST]);
                __builtin_amdgcn_s_setprio(3);
                mfma6(a0, b0, v0 % NH);
                __builtin_amdgcn_s_setprio(1);
                read_a(a1, g, cc1, v1 / NH);
                read_b(b1, st[s1 % F2_NST]);
                // -- conversions of this phase (behind the MFMAs just issued)
                __builtin_amdgcn_s_setprio(0);
                if (ph == P0_0 + CD || ph == P0_1 + CD) conv_consts();
#pragma unroll
                for (int q = 0; q < F2_NFV; ++q) {
                    if (ph >= P0_0 + CD && ph <= PL_0 && q / K_0 == ph - P0_0 - CD) convert_slice(q, 1);
                    if (ph >= P0_1 + CD && ph <= PL_1 && q / K_1 == ph - P0_1 - CD) convert_slice(q, 0);
                }
                __builtin_amdgcn_s_setprio(3);
                mfma6(a1, b1, v1 % NH);
                __builtin_amdgcn_s_setprio(1);
                // -- A fragments of the next phase's first step (read across the barrier: its footprint is complete)
                if (ph + 1 < NPH) read_a(a0, g, (s0 + 2) / NT, ((s0 + 2) % NT) / NH);
                else read_a(a0, gx1, 0, 0);
                // weights of phase ph + 1 were requested one phase ago: allow exactly this phase's own VMEM operations
                if (nld == 0) wait_vmcnt<2>();
                else if (nld == F2_NFV) wait_vmcnt<2 + F2_NFV>();
                else wait_vmcnt<2 + F2_NFV + NCHLD>();
                __builtin_amdgcn_s_barrier();
            }
            // rotate the stage roles by NSTEP steps
            if (NSTEP % F2_NST != 0) {
                int r[F2_NST];
#pragma unroll
                for (int i = 0; i < F2_NST; ++i) r[i] = st[i];
#pragma unroll
                for (int i = 0; i < F2_NST; ++i) st[i] = __builtin_amdgcn_readfirstlane(r[(i + NSTEP) % F2_NST]);   // (uniform: keep it scalar)
            }
        }
        // ---- tile complete: epilogue parameters through an opaque pointer (kept out of the main loop's registers)
        __builtin_amdgcn_s_setprio(0);
        {
            KArg q = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(q));                                  // opaque: these loads cannot be hoisted out of the epilogue
            EpiArgs e;
            e.bias = q->bias; e.ps = q->ps; e.pt = q->pt; e.res = q->res; e.out = q->out;
            e.M = q->M; e.Cout = q->Cout; e.act = q->act; e.pp = q->pp; e.poolkind = q->poolkind;
            if (TR) {
                epilogue_tr(e, acc0, acc1, (long long)tile * BM + wv * 32 + li, n0, lh);
                if (NH == 2) epilogue_tr(e, acc2, acc3, (long long)tile * BM + wv * 32 + li, n0 + BN, lh);
            } else {
                epilogue_tile(e, acc0, (long long)tile * BM + wv * 32, n0 + li, lh);
                epilogue_tile(e, acc1, (long long)tile * BM + wv * 32, n0 + 32 + li, lh);
                if (NH == 2) {
                    epilogue_tile(e, acc2, (long long)tile * BM + wv * 32, n0 + BN + li, lh);
                    epilogue_tile(e, acc3, (long long)tile * BM + wv * 32, n0 + BN + 32 + li, lh);
                }
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

template <int KH, int KW>
void launch_fp2_shape(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused, int nh) {
    if constexpr (KH * KW >= 8) {
#define ISS_F2_LAUNCH(...) hipLaunchKernelGGL((conv_x3_fp2_kernel<__VA_ARGS__>), grid, dim3(256), 0, st, a)
        if (fused) {
            if (tr) ISS_F2_LAUNCH(KH, KW, false, true, true, 1); else ISS_F2_LAUNCH(KH, KW, false, false, true, 1);
            return;
        }
        if constexpr (KH == 3 && KW == 3) {
            if (nh == 2) {
                if (padded && tr) ISS_F2_LAUNCH(KH, KW, true, true, false, 2);
                else if (padded) ISS_F2_LAUNCH(KH, KW, true, false, false, 2);
                else if (tr) ISS_F2_LAUNCH(KH, KW, false, true, false, 2);
                else ISS_F2_LAUNCH(KH, KW, false, false, false, 2);
                return;
            }
        }
        if (padded && tr) ISS_F2_LAUNCH(KH, KW, true, true, false, 1);
        else if (padded) ISS_F2_LAUNCH(KH, KW, true, false, false, 1);
        else if (tr) ISS_F2_LAUNCH(KH, KW, false, true, false, 1);
        else ISS_F2_LAUNCH(KH, KW, false, false, false, 1);
#undef ISS_F2_LAUNCH
    }
}

}  // namespace issk
