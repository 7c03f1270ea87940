// microbenchmark 2: 4 consumer waves (12 ds_read_b128 + 12 MFMA per iteration) + 1 producer wave that does the
// global loads, f32->bf16 hi/lo split and LDS writes a conv tap needs; one barrier per iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int PROD, int NBAR>
__global__ __launch_bounds__(320, 2) void k(float* out, const float4* src, int iters) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[38 * 1024];   // 76 KB -> 2 blocks / CU
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 38 * 1024; i += 320) lds[i] = (uint16_t)(0x3c00 + (i & 7));
    __syncthreads();
    if (wv < 4) {
        floatx16 acc0, acc1;
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
        const int base = ((wv * 32 + li) * 80 + lh * 16) / 2;
        bf16x8 f[12];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r) f[r] = *reinterpret_cast<const bf16x8*>(&lds[base + ((it + r) & 15) * 640 + r * 16]);
#pragma unroll
            for (int m = 0; m < 12; m += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m], f[(m + 5) % 12], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m + 1], f[(m + 6) % 12], acc1, 0, 0, 0);
            }
            for (int b = 0; b < NBAR; ++b) __syncthreads();
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        const float4* p = src + (size_t)blockIdx.x * 4096 + lane;
        for (int it = 0; it < iters; ++it) {
            if (PROD) {
                float4 v[11];
#pragma unroll
                for (int q = 0; q < 11; ++q) v[q] = p[((it * 11 + q) & 63) * 64];
#pragma unroll
                for (int q = 0; q < 3; ++q) {           // footprint slice: split + 2 x ds_write_b64
                    bf16x4 h, l;
                    h[0] = (__bf16)v[q].x; h[1] = (__bf16)v[q].y; h[2] = (__bf16)v[q].z; h[3] = (__bf16)v[q].w;
                    l[0] = (__bf16)(v[q].x - (float)h[0]); l[1] = (__bf16)(v[q].y - (float)h[1]);
                    l[2] = (__bf16)(v[q].z - (float)h[2]); l[3] = (__bf16)(v[q].w - (float)h[3]);
                    *reinterpret_cast<bf16x4*>(&lds[20480 + (q * 64 + lane) * 4]) = h;
                    *reinterpret_cast<bf16x4*>(&lds[24576 + (q * 64 + lane) * 4]) = l;
                }
#pragma unroll
                for (int q = 3; q < 11; ++q)             // B tile: 8 x ds_write_b128
                    *reinterpret_cast<float4*>(&lds[28672 + ((q - 3) * 64 + lane) * 8]) = v[q];
            }
            for (int b = 0; b < NBAR; ++b) __syncthreads();
        }
    }
}

template <int PROD, int NBAR>
void run(const char* name, float* d, float4* src) {
    const int iters = 4000, blocks = 512;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<PROD, NBAR>), dim3(blocks), dim3(320), 0, 0, d, src, 10);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PROD, NBAR>), dim3(blocks), dim3(320), 0, 0, d, src, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    printf("%-52s %8.3f ms  %7.1f TFLOP/s executed (%.1f%% of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    float4* src; (void)hipMalloc(&src, (size_t)512 * 4096 * 16 + 4096 * 16); (void)hipMemset(src, 0, (size_t)512 * 4096 * 16 + 4096 * 16);
    run<0, 1>("4 consumers + idle 5th wave, 1 barrier/iter", d, src);
    run<1, 1>("4 consumers + producer wave, 1 barrier/iter", d, src);
    run<1, 2>("4 consumers + producer wave, 2 barriers/iter", d, src);
    return 0;
}
