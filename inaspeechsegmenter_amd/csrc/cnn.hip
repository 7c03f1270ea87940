// Small-CNN / ResNet engine for gfx950: executes a flat op program (include/iss.h) on
// batches of 20 ms slots (or x-vector windows).
//
// Replaces, for the reference: segmenter.py:76-88 `_get_patches` (never materialised here:
// the 68 x nmel window is gathered straight from the resident (T,24) log-mel inside the
// first conv's operand loader, z-normalised on the fly), segmenter.py:156-163 (gather +
// `keras predict`) and vbx_segmenter.py:262-266 (onnxruntime ResNet-101, resnet.py:78-135).
//
// Kernels
//   patch_stats_kernel   per-slot mean / population-std / finite flag     (segmenter.py:82,86)
//   conv_igemm_kernel    conv2d + dense as implicit GEMM on v_mfma_f32_32x32x2_f32
//                        (exact f32, 157 TFLOP/s peak); fused bias, residual add,
//                        activation, post-activation scale/shift (BatchNorm), NHWC in/out
//   pool_kernel          max / average pooling, NHWC
//   softmax_kernel       softmax over channels
//   statpool_kernel      mean || std over time                           (resnet.py:123-127)
#include "iss_internal.h"
#include <cmath>
#include <cstring>
#include <algorithm>

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;      // GEMM rows (output pixels) per workgroup
constexpr int BN = 64;       // GEMM cols (output channels) per workgroup
constexpr int BK = 16;       // k-tile
constexpr int LDK = BK + 4;  // padded LDS row (floats): conflict-free ds_read_b128

struct ConvArgs {
    const float* in;
    const float* w;          // [Cout][Kpad]
    const float* bias;       // [Cout] or null
    const float* ps;         // post-activation scale [Cout] or null
    const float* pt;         // post-activation shift
    const float* res;        // residual, same shape as out, or null
    float* out;
    const int32_t* ktab;     // [Kpad] x {delta, (ky<<16)|kx}
    const int32_t* win_row;  // PATCH mode
    const float* stats;      // PATCH mode: {mean, std} per sample
    const uint8_t* finite;   // PATCH mode
    long long M;             // samples * Ho * Wo
    long long img_stride;    // floats per input sample
    int H, W, Cin, Ho, Wo, Cout;
    int sh, sw, pt_, pl_;
    int row_stride, pix_stride;
    int act, Kpad, mode;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    if (act == 3) return tanhf(v);
    return v;
}

// ------------------------------------------------------------------------------------------
// Implicit-GEMM convolution.  C[m][n] = sum_k A[m][k] * Wt[n][k],
//   m = (sample, oy, ox), n = cout, k = (ky, kx, cin)   (NHWC, weights [Cout][kh*kw*Cin]).
// 256 threads = 4 wavefronts; wave w owns rows [32w, 32w+32) x 64 cols = two 32x32 MFMA tiles.
// MODE: 0 = NHWC with Cin % 4 == 0 (float4 gathers), 1 = NHWC scalar gathers, 2 = z-normed patch.
template <int MODE>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDK];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDK];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-thread gather bookkeeping: this thread fills (row r, k4) and (row r+64, k4) of sA
    const int k4 = tid & 3;
    const int lr = tid >> 2;
    long long base[2];
    int iy0[2], ix0[2];
    float mean[2] = {0.f, 0.f}, sd[2] = {1.f, 1.f};
    bool ok[2];
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long long m = m0 + lr + 64 * j;
        ok[j] = m < p.M;
        const long long mm = ok[j] ? m : 0;
        const int b = (int)(mm / hw);
        const int rem = (int)(mm - (long long)b * hw);
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        iy0[j] = oy * p.sh - p.pt_;
        ix0[j] = ox * p.sw - p.pl_;
        if (MODE == 2) {
            base[j] = (long long)p.win_row[b] * 24 + (long long)iy0[j] * 24 + ix0[j];
            mean[j] = p.stats[2 * b];
            sd[j] = p.stats[2 * b + 1];
            ok[j] = ok[j] && p.finite[b];
        } else {
            base[j] = (long long)b * p.img_stride + (long long)iy0[j] * p.row_stride + (long long)ix0[j] * p.pix_stride;
        }
    }
    const int bn = n0 + lr;                       // weight row this thread stages
    const bool bok = bn < p.Cout;
    const float* wrow = p.w + (size_t)(bok ? bn : 0) * p.Kpad + k4 * 4;

    float4 ra[2], rb;
    auto gather = [&](int kt) {
        const int kbase = kt * BK + k4 * 4;
        if (MODE == 0) {
            const int2 e = reinterpret_cast<const int2*>(p.ktab)[kbase];
            const int ky = e.y >> 16, kx = e.y & 0xffff;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                if (ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    ra[j] = *reinterpret_cast<const float4*>(p.in + base[j] + e.x);
                else
                    ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int2 e = reinterpret_cast<const int2*>(p.ktab)[kbase + q];
                    const int ky = e.y >> 16, kx = e.y & 0xffff;
                    const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                    float x = 0.f;
                    if (ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                        x = p.in[base[j] + e.x];
                        if (MODE == 2) x = (x - mean[j]) / sd[j];
                    }
                    v[q] = x;
                }
                ra[j] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        rb = bok ? *reinterpret_cast<const float4*>(wrow + (size_t)kt * BK) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&sA[buf][lr * LDK + k4 * 4]) = ra[0];
        *reinterpret_cast<float4*>(&sA[buf][(lr + 64) * LDK + k4 * 4]) = ra[1];
        *reinterpret_cast<float4*>(&sB[buf][lr * LDK + k4 * 4]) = rb;
    };

    floatx16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    const int nk = p.Kpad / BK;
    gather(0);
    stage(0);
    __syncthreads();

    const int li = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gather(kt + 1);
        const float* a_s = &sA[cur][(wv * 32 + li) * LDK + lh * 4];
        const float* b_s = &sB[cur][li * LDK + lh * 4];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(a_s + g * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(b_s + g * 8);
            const float4 b1 = *reinterpret_cast<const float4*>(b_s + 32 * LDK + g * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
        }
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = n0 + t * 32 + li;
        if (n >= p.Cout) continue;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float s = p.ps ? p.ps[n] : 1.f;
        const float sh = p.pt ? p.pt[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const long long m = m0 + wv * 32 + row;
            if (m >= p.M) continue;
            float v = (t == 0 ? acc0[r] : acc1[r]) + bias;
            const size_t o = (size_t)m * p.Cout + n;
            if (p.res) v += p.res[o];
            v = apply_act(v, p.act);
            if (p.ps) v = v * s + sh;
            p.out[o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Per-slot statistics of the 68 x h log-mel window: mean, population std, finite flag.
// One wavefront per slot.  segmenter.py:82 (np.mean / np.std over the flattened window) and
// :86 (finite = all(isfinite(normalised))).
__global__ __launch_bounds__(256) void patch_stats_kernel(const float* __restrict__ mspec,
                                                          const int32_t* __restrict__ win_row, int n, int h,
                                                          float* __restrict__ stats, uint8_t* __restrict__ finite) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n) return;
    const float* src = mspec + (size_t)win_row[b] * 24;
    const int cnt = 68 * h;
    double s = 0.0;
    for (int e = lane; e < cnt; e += 64) { const int r = e / h, c = e - r * h; s += (double)src[r * 24 + c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const double mean_d = s / cnt;
    double q = 0.0;
    for (int e = lane; e < cnt; e += 64) {
        const int r = e / h, c = e - r * h;
        const double d = (double)src[r * 24 + c] - mean_d;
        q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float meanf = (float)mean_d;
    const float sdf = (float)sqrt(q / cnt);
    int bad = 0;
    for (int e = lane; e < cnt; e += 64) {
        const int r = e / h, c = e - r * h;
        const float v = (src[r * 24 + c] - meanf) / sdf;
        bad |= !isfinite(v);
    }
    bad = __any(bad);
    if (lane == 0) { stats[2 * b] = meanf; stats[2 * b + 1] = sdf; finite[b] = bad ? 0 : 1; }
}

// NHWC pooling; one thread per (sample, oy, ox, c)
__global__ void pool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int H, int W,
                            int C, int Ho, int Wo, int kh, int kw, int sh, int sw, int pt, int pl, int kind) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const long long b = t / Ho;
    float acc = kind == 0 ? -INFINITY : 0.f;
    int cnt = 0;
    for (int ky = 0; ky < kh; ++ky) {
        const int iy = oy * sh - pt + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < kw; ++kx) {
            const int ix = ox * sw - pl + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float v = in[((b * H + iy) * W + ix) * C + c];
            acc = kind == 0 ? fmaxf(acc, v) : acc + v;
            ++cnt;
        }
    }
    out[idx] = kind == 0 ? acc : acc / (float)(kh * kw);
    (void)cnt;
}

__global__ void softmax_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int C) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* x = in + r * C;
    float mx = x[0];
    for (int i = 1; i < C; ++i) mx = fmaxf(mx, x[i]);
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += expf(x[i] - mx);
    for (int i = 0; i < C; ++i) out[r * C + i] = expf(x[i] - mx) / s;
}

// mean || std over W (time) for every (h, c); out[(c*H + h)] and out[C*H + c*H + h]  (resnet.py:123-127)
__global__ void statpool_kernel(const float* __restrict__ in, float* __restrict__ out, long long total, int H, int W, int C) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long t = idx / C;
    const int h = (int)(t % H);
    const long long b = t / H;
    const float* src = in + ((b * H + h) * (long long)W) * C + c;
    float s = 0.f, q = 0.f;
    for (int w = 0; w < W; ++w) { const float v = src[(long long)w * C]; s += v; q += v * v; }
    const float mean = s / W, meansq = q / W;
    float* o = out + b * (2LL * C * H);
    o[c * H + h] = mean;
    o[(long long)C * H + c * H + h] = sqrtf(meansq - mean * mean + 1e-10f);
}

__global__ void fill_half_kernel(float* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.5f;
}

// rows whose window was not finite get 0.5 everywhere (segmenter.py:175)
__global__ void mask_probs_kernel(float* probs, const uint8_t* finite, long long n, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * C && !finite[i / C]) probs[i] = 0.5f;
}

inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

}  // namespace

// ============================================================================ host side
int iss_cnn_free(iss_ctx* c, int id) {
    if (!c || id < 0 || id >= ISS_MAX_NETS) return ISS_EINVAL;
    IssNet& n = c->nets[id];
    if (n.d_blob) (void)hipFree(n.d_blob);
    if (n.d_ktab) (void)hipFree(n.d_ktab);
    n = IssNet();
    return ISS_OK;
}

extern "C" int iss_cnn_load(iss_ctx* c, int id, const int32_t* prog, int32_t nrows, const float* blob,
                            int64_t blob_floats, int32_t nbuf, const int64_t* buf_elems, int32_t in_h, int32_t in_w,
                            int32_t in_c, int32_t out_dim) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || !prog || nrows <= 0 || !blob || blob_floats <= 0 || nbuf <= 0 || !buf_elems)
        return iss_fail(c, ISS_EINVAL, "iss_cnn_load: bad argument");
    ISS_HIP(c, hipSetDevice(c->device));
    iss_cnn_free(c, id);
    IssNet& n = c->nets[id];
    n.prog.assign(prog, prog + (size_t)nrows * ISS_PROG_COLS);
    n.nrows = nrows; n.nbuf = nbuf; n.buf_elems.assign(buf_elems, buf_elems + nbuf);
    n.in_h = in_h; n.in_w = in_w; n.in_c = in_c; n.out_dim = out_dim;
    n.kpad.assign(nrows, 0); n.ktab_off.assign(nrows, -1);
    std::vector<int32_t> ktab;
    double flops = 0;
    for (int r = 0; r < nrows; ++r) {
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        auto bad = [&](const char* what) { return iss_fail(c, ISS_EINVAL, "iss_cnn_load: row %d: %s", r, what); };
        if (R[ISS_C_IN] != ISS_BUF_INPUT && (R[ISS_C_IN] < 0 || R[ISS_C_IN] >= nbuf)) return bad("IN buffer id");
        if (R[ISS_C_OUT] < 0 || R[ISS_C_OUT] >= nbuf) return bad("OUT buffer id");
        if (R[ISS_C_OP] == ISS_OP_CONV) {
            const int K = R[ISS_C_KH] * R[ISS_C_KW] * R[ISS_C_CIN];
            const int Kpad = roundup(K, BK);
            if (K <= 0 || R[ISS_C_COUT] <= 0) return bad("conv shape");
            if (R[ISS_C_WOFF] < 0 || (int64_t)R[ISS_C_WOFF] + (int64_t)R[ISS_C_COUT] * Kpad > blob_floats)
                return bad("weight offset outside blob (weights must be [Cout][roundup16(K)])");
            if (R[ISS_C_RES] >= nbuf) return bad("RES buffer id");
            if (R[ISS_C_KH] > 32767 || R[ISS_C_KW] > 32767) return bad("kernel too large");
            const bool patch = R[ISS_C_INMODE] == 1;
            if (patch && (R[ISS_C_CIN] != 1 || R[ISS_C_H] != 68 || R[ISS_C_W] > 24)) return bad("patch-mode conv must read (68, <=24, 1)");
            const int rs = patch ? 24 : R[ISS_C_W] * R[ISS_C_CIN], ps = patch ? 1 : R[ISS_C_CIN];
            n.kpad[r] = Kpad; n.ktab_off[r] = (int64_t)ktab.size();
            for (int k = 0; k < Kpad; ++k) {
                if (k < K) {
                    const int cc = k % R[ISS_C_CIN], kk = k / R[ISS_C_CIN];
                    const int kx = kk % R[ISS_C_KW], ky = kk / R[ISS_C_KW];
                    ktab.push_back(ky * rs + kx * ps + cc);
                    ktab.push_back((ky << 16) | kx);
                } else {                              // K padding: force the bounds test to fail -> zeros
                    ktab.push_back(0);
                    ktab.push_back((0x7fff << 16) | 0x7fff);
                }
            }
            flops += 2.0 * K * R[ISS_C_COUT] * R[ISS_C_HO] * R[ISS_C_WO];
        } else if (R[ISS_C_OP] != ISS_OP_POOL && R[ISS_C_OP] != ISS_OP_SOFTMAX && R[ISS_C_OP] != ISS_OP_STATPOOL) {
            return bad("unknown op");
        }
    }
    n.flops_per_sample = flops; n.blob_floats = blob_floats;
    ISS_HIP(c, hipMalloc((void**)&n.d_blob, (size_t)blob_floats * sizeof(float)));
    ISS_HIP(c, hipMemcpy(n.d_blob, blob, (size_t)blob_floats * sizeof(float), hipMemcpyHostToDevice));
    if (!ktab.empty()) {
        ISS_HIP(c, hipMalloc((void**)&n.d_ktab, ktab.size() * sizeof(int32_t)));
        ISS_HIP(c, hipMemcpy(n.d_ktab, ktab.data(), ktab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    n.loaded = true;
    return ISS_OK;
}

extern "C" int iss_cnn_flops(iss_ctx* c, int id, double* f) {
    if (!c || id < 0 || id >= ISS_MAX_NETS || !f) return ISS_EINVAL;
    if (!c->nets[id].loaded) return iss_fail(c, ISS_ESTATE, "net %d not loaded", id);
    *f = c->nets[id].flops_per_sample;
    return ISS_OK;
}

namespace {

// Run the op program on `bc` samples.  src: PATCH mode uses (d_winrow + s0, stats, finite),
// otherwise `d_input` is an NHWC batch.  The result is left in act[last OUT].
int run_program(iss_ctx* c, IssNet& n, int bc, const int32_t* d_winrow, const float* d_stats,
                const uint8_t* d_fin, const float* d_input, float** result) {
    for (int r = 0; r < n.nrows; ++r) {
        const int32_t* R = &n.prog[(size_t)r * ISS_PROG_COLS];
        const float* in = R[ISS_C_IN] == ISS_BUF_INPUT ? d_input : (const float*)c->act[R[ISS_C_IN]].p;
        float* out = (float*)c->act[R[ISS_C_OUT]].p;
        const int op = R[ISS_C_OP];
        if (op == ISS_OP_CONV) {
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.in = in;
            a.w = n.d_blob + R[ISS_C_WOFF];
            a.bias = R[ISS_C_BOFF] >= 0 ? n.d_blob + R[ISS_C_BOFF] : nullptr;
            a.ps = R[ISS_C_PSOFF] >= 0 ? n.d_blob + R[ISS_C_PSOFF] : nullptr;
            a.pt = R[ISS_C_PTOFF] >= 0 ? n.d_blob + R[ISS_C_PTOFF] : nullptr;
            a.res = R[ISS_C_RES] >= 0 ? (const float*)c->act[R[ISS_C_RES]].p : nullptr;
            a.out = out;
            a.ktab = n.d_ktab + n.ktab_off[r];
            a.H = R[ISS_C_H]; a.W = R[ISS_C_W]; a.Cin = R[ISS_C_CIN];
            a.Ho = R[ISS_C_HO]; a.Wo = R[ISS_C_WO]; a.Cout = R[ISS_C_COUT];
            a.sh = R[ISS_C_SH]; a.sw = R[ISS_C_SW]; a.pt_ = R[ISS_C_PT]; a.pl_ = R[ISS_C_PL];
            a.act = R[ISS_C_ACT]; a.Kpad = n.kpad[r];
            a.M = (long long)bc * a.Ho * a.Wo;
            const bool patch = R[ISS_C_INMODE] == 1;
            a.mode = patch ? 2 : ((a.Cin % 4 == 0) ? 0 : 1);
            if (patch) {
                if (!d_winrow) return iss_fail(c, ISS_ESTATE, "patch-mode network run without a window list");
                a.in = (const float*)c->mspec.p; a.win_row = d_winrow; a.stats = d_stats; a.finite = d_fin;
                a.row_stride = 24; a.pix_stride = 1; a.img_stride = 0;
            } else {
                if (!in) return iss_fail(c, ISS_ESTATE, "network input missing");
                a.row_stride = a.W * a.Cin; a.pix_stride = a.Cin; a.img_stride = (long long)a.H * a.W * a.Cin;
            }
            dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
            const double fl = 2.0 * R[ISS_C_KH] * R[ISS_C_KW] * a.Cin * (double)a.Cout * (double)a.M;
            iss_prof_begin(c, 0, fl);
            if (a.mode == 0) hipLaunchKernelGGL(conv_igemm_kernel<0>, grid, dim3(256), 0, c->stream, a);
            else if (a.mode == 1) hipLaunchKernelGGL(conv_igemm_kernel<1>, grid, dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL(conv_igemm_kernel<2>, grid, dim3(256), 0, c->stream, a);
            iss_prof_end(c);
        } else if (op == ISS_OP_POOL) {
            const long long total = (long long)bc * R[ISS_C_HO] * R[ISS_C_WO] * R[ISS_C_CIN];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               total, R[ISS_C_H], R[ISS_C_W], R[ISS_C_CIN], R[ISS_C_HO], R[ISS_C_WO], R[ISS_C_KH],
                               R[ISS_C_KW], R[ISS_C_SH], R[ISS_C_SW], R[ISS_C_PT], R[ISS_C_PL], R[ISS_C_POOLKIND]);
            iss_prof_end(c);
        } else if (op == ISS_OP_SOFTMAX) {
            const long long rows = (long long)bc * R[ISS_C_H] * R[ISS_C_W];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               rows, R[ISS_C_CIN]);
            iss_prof_end(c);
        } else if (op == ISS_OP_STATPOOL) {
            const long long total = (long long)bc * R[ISS_C_H] * R[ISS_C_CIN];
            iss_prof_begin(c, 2, 0);
            hipLaunchKernelGGL(statpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, in, out,
                               total, R[ISS_C_H], R[ISS_C_W], R[ISS_C_CIN]);
            iss_prof_end(c);
        }
        ISS_HIP(c, hipGetLastError());
        *result = out;
    }
    return ISS_OK;
}

int plan_chunk(iss_ctx* c, IssNet& n, int total, int* bc_out) {
    int64_t per = 0;
    for (auto e : n.buf_elems) per += e;
    int64_t bc = (int64_t)(c->ws_limit / (uint64_t)(per * sizeof(float)));
    if (bc < 1) bc = 1;
    if (bc > total) bc = total;
    if (bc > 8) bc -= bc % 8;
    // keep every per-buffer float index below 2^31 (ConvArgs uses 64-bit bases, pool uses 64-bit too; this is a sanity cap)
    if ((int)c->act.size() < n.nbuf) c->act.resize(n.nbuf);
    for (int i = 0; i < n.nbuf; ++i) {
        int rc = iss_reserve(c, c->act[i], (size_t)bc * n.buf_elems[i] * sizeof(float));
        if (rc) return rc;
    }
    *bc_out = (int)bc;
    return ISS_OK;
}

}  // namespace

extern "C" int iss_cnn_probs(iss_ctx* c, int id, const int32_t* win_row, int32_t nslots, float* probs_out,
                             uint8_t* finite_out) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || nslots < 0 || (nslots > 0 && (!win_row || !probs_out || !finite_out)))
        return iss_fail(c, ISS_EINVAL, "iss_cnn_probs: bad argument");
    IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "iss_cnn_probs: net %d not loaded", id);
    if (!c->have_feats) return iss_fail(c, ISS_ESTATE, "iss_cnn_probs: no mel spectrogram resident");
    if (n.in_h != 68 || n.in_c != 1 || n.in_w > 24) return iss_fail(c, ISS_EINVAL, "net %d is not a (68,h,1) patch network", id);
    if (nslots == 0) return ISS_OK;
    for (int i = 0; i < nslots; ++i)
        if (win_row[i] < 0 || win_row[i] + 68 > c->T)
            return iss_fail(c, ISS_EINVAL, "iss_cnn_probs: window %d (row %d) outside the %d resident frames", i, win_row[i], c->T);
    ISS_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = iss_reserve(c, c->d_winrow, (size_t)nslots * 4))) return rc;
    if ((rc = iss_reserve(c, c->d_stats, (size_t)nslots * 8))) return rc;
    if ((rc = iss_reserve(c, c->d_finite, (size_t)nslots))) return rc;
    if ((rc = iss_reserve(c, c->d_out, (size_t)nslots * n.out_dim * 4))) return rc;
    int bc = 0;
    if ((rc = plan_chunk(c, n, nslots, &bc))) return rc;
    ISS_HIP(c, hipMemcpyAsync(c->d_winrow.p, win_row, (size_t)nslots * 4, hipMemcpyHostToDevice, c->stream));
    iss_prof_begin(c, 2, 0);
    hipLaunchKernelGGL(patch_stats_kernel, dim3((nslots + 3) / 4), dim3(256), 0, c->stream, (const float*)c->mspec.p,
                       (const int32_t*)c->d_winrow.p, nslots, n.in_w, (float*)c->d_stats.p, (uint8_t*)c->d_finite.p);
    iss_prof_end(c);
    ISS_HIP(c, hipGetLastError());
    for (int s0 = 0; s0 < nslots; s0 += bc) {
        const int cur = std::min(bc, nslots - s0);
        float* res = nullptr;
        rc = run_program(c, n, cur, (const int32_t*)c->d_winrow.p + s0, (const float*)c->d_stats.p + 2 * (size_t)s0,
                         (const uint8_t*)c->d_finite.p + s0, nullptr, &res);
        if (rc) return rc;
        ISS_HIP(c, hipMemcpyAsync((float*)c->d_out.p + (size_t)s0 * n.out_dim, res, (size_t)cur * n.out_dim * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    const long long tot = (long long)nslots * n.out_dim;
    hipLaunchKernelGGL(mask_probs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream,
                       (float*)c->d_out.p, (const uint8_t*)c->d_finite.p, (long long)nslots, n.out_dim);
    ISS_HIP(c, hipGetLastError());
    ISS_HIP(c, hipMemcpyAsync(probs_out, c->d_out.p, (size_t)tot * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipMemcpyAsync(finite_out, c->d_finite.p, (size_t)nslots, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}

extern "C" int iss_cnn_forward(iss_ctx* c, int id, const float* x, int32_t nsamp, float* out) {
    if (!c) return ISS_EINVAL;
    if (id < 0 || id >= ISS_MAX_NETS || nsamp < 0 || (nsamp > 0 && (!x || !out)))
        return iss_fail(c, ISS_EINVAL, "iss_cnn_forward: bad argument");
    IssNet& n = c->nets[id];
    if (!n.loaded) return iss_fail(c, ISS_ESTATE, "iss_cnn_forward: net %d not loaded", id);
    for (int r = 0; r < n.nrows; ++r)
        if (n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_OP] == ISS_OP_CONV && n.prog[(size_t)r * ISS_PROG_COLS + ISS_C_INMODE] == 1)
            return iss_fail(c, ISS_EINVAL, "iss_cnn_forward: net %d reads mspec patches; use iss_cnn_probs", id);
    if (nsamp == 0) return ISS_OK;
    ISS_HIP(c, hipSetDevice(c->device));
    const size_t in_elems = (size_t)n.in_h * n.in_w * n.in_c;
    int rc, bc = 0;
    if ((rc = plan_chunk(c, n, nsamp, &bc))) return rc;
    if ((rc = iss_reserve(c, c->d_in, (size_t)bc * in_elems * 4))) return rc;
    if ((rc = iss_reserve(c, c->d_out, (size_t)nsamp * n.out_dim * 4))) return rc;
    for (int s0 = 0; s0 < nsamp; s0 += bc) {
        const int cur = std::min(bc, nsamp - s0);
        ISS_HIP(c, hipMemcpyAsync(c->d_in.p, x + (size_t)s0 * in_elems, (size_t)cur * in_elems * 4, hipMemcpyHostToDevice, c->stream));
        float* res = nullptr;
        rc = run_program(c, n, cur, nullptr, nullptr, nullptr, (const float*)c->d_in.p, &res);
        if (rc) return rc;
        ISS_HIP(c, hipMemcpyAsync((float*)c->d_out.p + (size_t)s0 * n.out_dim, res, (size_t)cur * n.out_dim * 4,
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    ISS_HIP(c, hipMemcpyAsync(out, c->d_out.p, (size_t)nsamp * n.out_dim * 4, hipMemcpyDeviceToHost, c->stream));
    ISS_HIP(c, hipStreamSynchronize(c->stream));
    iss_prof_collect(c);
    return ISS_OK;
}
