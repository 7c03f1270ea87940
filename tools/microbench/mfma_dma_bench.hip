// microbenchmark 3: 4 MFMA waves (12 ds_read_b128 + 12 MFMA / iteration) that also feed their own LDS through
// LDS-DMA (global_load_lds_dwordx4, inline asm, counted vmcnt, raw s_barrier): 3 x 1 KiB per wave per iteration,
// fetch distance 2 iterations.  Checks the DMA'd bytes at the end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NDMA, int DIST>
__global__ __launch_bounds__(256, 2) void k(float* out, const uint4* src, int iters, unsigned* check) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[36 * 1024];   // 72 KB: 24 KB read area + 3 stages x 12 KB DMA area
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 36 * 1024; i += 256) lds[i] = (uint16_t)(0x3c00 + (i & 7));
    __syncthreads();
    floatx16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const int base = ((wv * 32 + li) * 80 + lh * 16) / 2;
    const unsigned lds_base = (unsigned)(size_t)(&lds[0]);             // LDS byte address of the array
    const uint4* p = src + (size_t)(blockIdx.x & 1) * 16384;   // L2-resident source (like conv weights)
    bf16x8 f[12];
    for (int it = 0; it < iters; ++it) {
        // DMA for iteration it + DIST into stage (it % 3): each wave 3 x 1 KiB
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            const unsigned dst = lds_base + 24576 + (unsigned)((it % 3) * 12288 + (wv * 3 + q) * 1024);
            glds16(p + ((it * 12 + wv * 3 + q) & 255) * 64 + lane, dst);
        }
#pragma unroll
        for (int r = 0; r < 12; ++r) f[r] = *reinterpret_cast<const bf16x8*>(&lds[base + ((it + r) & 15) * 640 + r * 16]);
#pragma unroll
        for (int m = 0; m < 12; m += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m], f[(m + 5) % 12], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[m + 1], f[(m + 6) % 12], acc1, 0, 0, 0);
        }
        // everything issued before the last DIST-1 iterations must have landed
        if (NDMA > 0) {
            if (DIST == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (NDMA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + tid] = s;
    if (blockIdx.x == 0 && NDMA > 0) {                                 // last iteration's stage, wave's first piece
        const int it = iters - 1;
        const unsigned* l32 = reinterpret_cast<const unsigned*>(&lds[(24576 + (it % 3) * 12288 + (wv * 3) * 1024) / 2]);
        check[tid * 4 + 0] = l32[lane * 4 + 0];
        check[tid * 4 + 1] = l32[lane * 4 + 1];
    }
}

template <int NDMA, int DIST>
void run(const char* name, float* d, uint4* src, unsigned* chk, const std::vector<uint32_t>& host) {
    const int iters = 4000, blocks = 512;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<NDMA, DIST>), dim3(blocks), dim3(256), 0, 0, d, src, 10, chk);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<NDMA, DIST>), dim3(blocks), dim3(256), 0, 0, d, src, iters, chk);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
    int bad = 0;
    if (NDMA > 0) {
        std::vector<unsigned> c(1024); (void)hipMemcpy(c.data(), chk, 4096, hipMemcpyDeviceToHost);
        const int it = iters - 1;
        for (int t = 0; t < 256; ++t) {
            const int wv = t >> 6, lane = t & 63;
            const size_t idx = ((size_t)(((it * 12 + wv * 3) & 255) * 64 + lane)) * 4;   // uint4 index * 4 words
            if (c[t * 4] != host[idx] || c[t * 4 + 1] != host[idx + 1]) ++bad;
        }
    }
    printf("%-46s %8.3f ms  %7.1f TFLOP/s executed (%.1f%% of 2500)  dma-check-bad=%d\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0, bad);
}
int main() {
    float* d; (void)hipMalloc(&d, 512 * 256 * 4);
    const size_t nsrc = (size_t)512 * 16384 + 16384;
    std::vector<uint32_t> host(nsrc * 4);
    for (size_t i = 0; i < host.size(); ++i) host[i] = (uint32_t)(i * 2654435761u);
    uint4* src; (void)hipMalloc(&src, nsrc * 16); (void)hipMemcpy(src, host.data(), nsrc * 16, hipMemcpyHostToDevice);
    unsigned* chk; (void)hipMalloc(&chk, 4096);
    run<0, 1>("no DMA, raw barrier", d, src, chk, host);
    run<3, 1>("3 x 1 KiB DMA / wave / iter, distance 1", d, src, chk, host);
    run<3, 2>("3 x 1 KiB DMA / wave / iter, distance 2", d, src, chk, host);
    run<2, 2>("2 x 1 KiB DMA / wave / iter, distance 2", d, src, chk, host);
    return 0;
}
