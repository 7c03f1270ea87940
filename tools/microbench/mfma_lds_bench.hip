// microbenchmark: what does the matrix pipe sustain when every 12 MFMAs need N ds_read_b128 ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NREAD, int STRIDE_B, bool BARRIER, int NACC>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[36 * 1024];   // 72 KB -> 2 blocks / CU
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 36 * 1024; i += 256) lds[i] = (uint16_t)(0x3c00 + (i & 7));
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const int base = ((wv * 32 + li) * STRIDE_B + lh * 16) / 2;        // element offset, row stride STRIDE_B bytes
    bf16x8 f[12];
    for (int r = 0; r < 12; ++r) f[r] = *reinterpret_cast<const bf16x8*>(&lds[base + r * 16]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < NREAD; ++r) f[r] = *reinterpret_cast<const bf16x8*>(&lds[base + ((it + r) & 15) * 640 + r * 16]);
#pragma unroll
        for (int m = 0; m < 12; ++m)
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m % 12], f[(m + 5) % 12], acc[m % NACC], 0, 0, 0);
        if (BARRIER) __syncthreads();
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NREAD, int STRIDE_B, bool BARRIER, int NACC>
void run(const char* name, float* d) {
    const int iters = 4000, blocks = 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NREAD, STRIDE_B, BARRIER, NACC>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NREAD, STRIDE_B, BARRIER, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s executed (%.1f%% of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}
int main() {
    float* d; hipMalloc(&d, 512 * 256 * 4);
    run<0, 80, false, 2>("0 reads, 2 acc", d);
    run<0, 80, false, 4>("0 reads, 4 acc", d);
    run<6, 80, false, 2>("6 reads/12 mfma stride 80B, 2 acc", d);
    run<12, 80, false, 2>("12 reads/12 mfma stride 80B, 2 acc", d);
    run<12, 80, true, 2>("12 reads/12 mfma stride 80B, 2 acc, barrier", d);
    run<12, 64, false, 2>("12 reads/12 mfma stride 64B (conflicts)", d);
    run<12, 80, false, 4>("12 reads/12 mfma stride 80B, 4 acc", d);
    return 0;
}
