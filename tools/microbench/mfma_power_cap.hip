// What MFMA rate does an MI355X of this pool SUSTAIN on v_mfma_f32_32x32x16_{f16,bf16} with every CU busy?  (The dense peak is
// 2.5 PFLOP/s at 2.4 GHz; the segmenter step's conv kernels all sit at matrix-pipe busy x clock ~ 117 %.GHz, i.e. ~1.25-1.37 PF executed.)
// Register-resident operands, 6 accumulators per wave, no memory traffic in the loop; operands random normal (as activations / weights
// are), zeros, or the hi | lo mix of a split operand; 1 or 2 waves per SIMD; ~0.2 s per case so that the power manager settles.
//   hipcc --offload-arch=gfx950 -O3 mfma_power_cap.hip -o mfma_power_cap && ./mfma_power_cap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool F16>
__global__ __launch_bounds__(512) void k(const uint4* src, float* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 a[2], b[3];
    a[0] = src[(t * 5 + 0) & 65535]; a[1] = src[(t * 5 + 1) & 65535];
    b[0] = src[(t * 5 + 2) & 65535]; b[1] = src[(t * 5 + 3) & 65535]; b[2] = src[(t * 5 + 4) & 65535];
    floatx16 c[2][3];
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 3; ++q) for (int i = 0; i < 16; ++i) c[r][q][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    if (F16) c[r][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[r]), __builtin_bit_cast(half8, b[q]), c[r][q], 0, 0, 0);
                    else c[r][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a[r]), __builtin_bit_cast(bf8, b[q]), c[r][q], 0, 0, 0);
                }
        // keep the accumulators bounded without leaving the matrix pipe idle for long: one cheap scale every 24 MFMAs on one register
        c[0][0][0] *= 0.5f;
    }
    float s = 0.f;
    for (int r = 0; r < 2; ++r) for (int q = 0; q < 3; ++q) for (int i = 0; i < 16; ++i) s += c[r][q][i];
    if (s == 12345.678f) out[t] = s;
}

static uint16_t f16bits(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static uint16_t bf16bits(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float gauss() { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

int main(int argc, char** argv) {
    const int N = 65536 * 8;                        // 16-bit elements
    std::vector<uint16_t> h(N);
    uint4* d; float* o;
    hipMalloc(&d, N * 2); hipMalloc(&o, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"random normal", "zeros", "relu(normal): half of A's values are zero", "lo parts (2^-11 of a normal)",
                           "relu'd A x B whose mantissas keep 5 bits (a short lo part)", "relu'd A x B whose mantissas keep 2 bits",
                           "A and B mantissas keep 5 bits"};
    for (int f16 = 1; f16 >= 0; --f16)
        for (int data = 0; data < 7; ++data)
            for (int waves = 4; waves <= 8; waves += 4) {
                for (int i = 0; i < N; ++i) {
                    float x = data == 1 ? 0.f : gauss();
                    if (data == 2 && (i / 8) % 5 < 2) x = x > 0 ? x : 0.f;       // the A registers
                    if (data == 3) x *= 4.8828125e-4f;
                    if (data >= 4 && (i / 8) % 5 < 2 && data != 6) x = x > 0 ? x : 0.f;
                    h[i] = f16 ? f16bits(x) : bf16bits(x);
                    // B registers (and A for case 6): clear the low mantissa bits (f16: 10 stored bits, bf16: 7)
                    if (data >= 4 && ((i / 8) % 5 >= 2 || data == 6)) {
                        const int keep = data == 5 ? 2 : 5, stored = f16 ? 10 : 7;
                        if (stored > keep) h[i] &= (uint16_t)~((1u << (stored - keep)) - 1u);
                    }
                }
                hipMemcpy(d, h.data(), N * 2, hipMemcpyHostToDevice);
                const int iters = 60000 * (argc > 1 ? atoi(argv[1]) : 1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (f16) hipLaunchKernelGGL(k<true>, dim3(256), dim3(waves * 64), 0, 0, d, o, iters);
                    else hipLaunchKernelGGL(k<false>, dim3(256), dim3(waves * 64), 0, 0, d, o, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    const double flops = 256.0 * waves * iters * 24.0 * 2.0 * 32 * 32 * 16;
                    if (rep == 1)
                        printf("%-5s %-44s %d waves/CU: %7.1f ms  %7.1f TFLOP/s executed = %.3f of 2500  (%.0f MHz-equivalent at 1024 flop/clk/SIMD)\n",
                               f16 ? "f16" : "bf16", names[data], waves, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500.0,
                               flops / (ms * 1e-3) / (256.0 * 4 * 1024) * 1e-6);
                }
            }
    return 0;
}
