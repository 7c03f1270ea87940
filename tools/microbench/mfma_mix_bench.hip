// microbenchmark: what does the matrix pipe sustain with the instruction mix of the split-operand (bf16x3) conv kernels?
// Per 12 MFMAs (two k16 steps of a 32 x 64 wave tile): NREAD ds_read_b128 (conflict-free), NVALU plain f32 VALU
// instructions on registers the MFMAs do not touch, NCVT v_cvt_pk_bf16_f32, optional ds_write_b64.  Interleaved between
// the MFMA pairs by sched_barrier exactly like conv_x3_ws_kernel.  2 x 256-thread workgroups per CU (two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 scratch/mfma_mix_bench.hip -o scratch/mfma_mix_bench && scratch/mfma_mix_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int NREAD, int NVALU, int NCVT, int NWRITE, int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[36 * 1024];   // 72 KB -> 2 blocks / CU
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 36 * 1024; i += 256) lds[i] = (uint16_t)(0x3c00 + (i & 7));
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const int base = ((wv * 32 + li) * 80 + lh * 16) / 2;              // element offset, 80-byte rows: conflict-free
    bf16x8 f[12];
    for (int r = 0; r < 12; ++r) f[r] = *reinterpret_cast<const bf16x8*>(&lds[base + r * 16]);
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 0.001f * (tid + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {                                    // six MFMA pairs
            __builtin_amdgcn_sched_barrier(0);
            acc[(2 * g) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(2 * g) % 12], f[(2 * g + 5) % 12], acc[(2 * g) % NACC], 0, 0, 0);
            acc[(2 * g + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(2 * g + 1) % 12], f[(2 * g + 6) % 12], acc[(2 * g + 1) % NACC], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = g * NREAD / 6; r < (g + 1) * NREAD / 6; ++r)
                f[(r + 3) % 12] = *reinterpret_cast<const bf16x8*>(&lds[base + ((it + r) & 15) * 640 + r * 16]);
#pragma unroll
            for (int v = g * NVALU / 6; v < (g + 1) * NVALU / 6; ++v) x[v & 7] = fmaf(x[v & 7], 1.0001f, x[(v + 3) & 7]);
#pragma unroll
            for (int v = g * NCVT / 6; v < (g + 1) * NCVT / 6; ++v) {
                bf16x2 h; h[0] = (__bf16)x[v & 7]; h[1] = (__bf16)x[(v + 1) & 7];
                x[(v + 2) & 7] += (float)h[0];
            }
#pragma unroll
            for (int w = g * NWRITE / 6; w < (g + 1) * NWRITE / 6; ++w)
                *reinterpret_cast<float2*>(&lds[(32 * 1024 + tid * 4 + w * 1024) & (36 * 1024 - 4)]) = make_float2(x[w & 7], x[(w + 1) & 7]);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NREAD, int NVALU, int NCVT, int NWRITE, int NACC>
void run(const char* name, float* d) {
    const int iters = 4000, blocks = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NREAD, NVALU, NCVT, NWRITE, NACC>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NREAD, NVALU, NCVT, NWRITE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    printf("%-70s %8.3f ms  %7.1f TFLOP/s executed (%.1f%% of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<0, 0, 0, 0, 2>("pure MFMA, 2 accumulators", d);
    run<0, 0, 0, 0, 4>("pure MFMA, 4 accumulators", d);
    run<12, 0, 0, 0, 2>("12 ds_read_b128 / 12 MFMA", d);
    run<8, 0, 0, 0, 4>("8 ds_read_b128 / 12 MFMA, 4 acc (64-row wave tile)", d);
    run<0, 12, 0, 0, 2>("12 VALU / 12 MFMA", d);
    run<0, 24, 0, 0, 2>("24 VALU / 12 MFMA", d);
    run<0, 36, 0, 0, 2>("36 VALU / 12 MFMA", d);
    run<12, 12, 0, 0, 2>("12 ds_read + 12 VALU / 12 MFMA", d);
    run<12, 24, 6, 0, 2>("12 ds_read + 24 VALU + 6 cvt / 12 MFMA", d);
    run<12, 24, 6, 2, 2>("12 ds_read + 24 VALU + 6 cvt + 2 ds_write_b64 / 12 MFMA (conv mix)", d);
    run<12, 36, 6, 2, 2>("12 ds_read + 36 VALU + 6 cvt + 2 ds_write_b64 / 12 MFMA", d);
    run<8, 12, 6, 2, 4>("8 ds_read + 12 VALU + 6 cvt + 2 ds_write / 12 MFMA, 4 acc", d);
    return 0;
}
