// microbenchmark: ONE wave per SIMD (256 threads, 512 registers per lane), 24 MFMA slots per iteration on 16 accumulators:
// what does an LDS fragment read / a VALU instruction cost when it sits in the shadow of an MFMA?  (conv_wq.h's stream.)
//   MODE 0: reads consumed by the next iteration's MFMAs (double-buffered fragment sets -> compiler lgkmcnt waits)
//   MODE 1: reads into registers no MFMA uses (kept alive at the end: no waits inside the loop)
//   MODE 2: like 0 with ds_read_b64
//   PAIR  : reads placed two per slot in NREAD / 2 slots instead of one per slot
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_slot_bench.hip -o /tmp/mfma_slot_bench && /tmp/mfma_slot_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int NREAD, int NVALU, int MODE, bool PAIR, int GAP>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[60 * 1024];   // 120 KB -> one block per CU
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 60 * 1024; i += 256) lds[i] = (uint16_t)(0x3c00 + (i & 7));
    __syncthreads();
    floatx16 acc[16];
    for (int a = 0; a < 16; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const int base = ((wv * 32 + li) * 80 + lh * 16) / 2;              // element offset, 80-byte rows: conflict-free
    bf16x8 f[2][12], spare[12];
    for (int s = 0; s < 2; ++s) for (int r = 0; r < 12; ++r) f[s][r] = *reinterpret_cast<const bf16x8*>(&lds[base + r * 16]);
    for (int r = 0; r < 12; ++r) spare[r] = f[0][r];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + 0.001f * (tid + i);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int cs = half, ns = half ^ 1;
            int rd = 0, va = 0;
#pragma unroll
            for (int s = 0; s < 24; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                acc[s % 16] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[cs][s % 8], f[cs][8 + (s % 4)], acc[s % 16], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int nr = PAIR ? ((s % 2 == 0 && rd < NREAD) ? 2 : 0) : ((s >= GAP && rd < NREAD && s - GAP < 24) ? 1 : 0);
#pragma unroll
                for (int j = 0; j < nr && rd < NREAD; ++j, ++rd) {
                    const int r = rd % 12;
                    const uint16_t* src = &lds[base + (((it + half) & 7) * 12 + r) * 40 * 64 / 2 % (56 * 1024) + r * 16];
                    if (MODE == 1) spare[r] = *reinterpret_cast<const bf16x8*>(src);
                    else if (MODE == 2) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(src); f[ns][r][0] = t[0]; f[ns][r][1] = t[1]; f[ns][r][2] = t[2]; f[ns][r][3] = t[3]; }
                    else f[ns][r] = *reinterpret_cast<const bf16x8*>(src);
                }
#pragma unroll
                for (int j = 0; j < (NVALU * (s + 1)) / 24 - (NVALU * s) / 24; ++j, ++va) x[va & 7] = fmaf(x[va & 7], 1.0001f, x[(va + 3) & 7]);
            }
        }
    }
    float sum = 0;
    for (int a = 0; a < 16; ++a) for (int i = 0; i < 16; ++i) sum += acc[a][i];
    for (int i = 0; i < 8; ++i) sum += x[i];
    for (int r = 0; r < 12; ++r) sum += (float)spare[r][0];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int NREAD, int NVALU, int MODE, bool PAIR, int GAP = 0>
void run(const char* name, float* d) {
    const int iters = 4000, blocks = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NREAD, NVALU, MODE, PAIR, GAP>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NREAD, NVALU, MODE, PAIR, GAP>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double flops = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("%-78s %8.3f ms  %7.1f TFLOP/s executed (%.1f%% of 2500)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 4);
    run<0, 0, 0, false>("pure MFMA (24 slots, 16 accumulators, one wave per SIMD)", d);
    run<6, 0, 0, false>("6 ds_read_b128 / 24 MFMA, one per slot, consumed next iteration", d);
    run<12, 0, 0, false>("12 ds_read_b128 / 24 MFMA, one per slot, consumed next iteration", d);
    run<24, 0, 0, false>("24 ds_read_b128 / 24 MFMA, one per slot, consumed next iteration", d);
    run<12, 0, 1, false>("12 ds_read_b128 / 24 MFMA, results unused (no waits in the loop)", d);
    run<12, 0, 2, false>("12 ds_read_b64 / 24 MFMA, consumed", d);
    run<12, 0, 0, true>("12 ds_read_b128 / 24 MFMA, two per slot in six slots", d);
    run<12, 0, 0, false, 12>("12 ds_read_b128 / 24 MFMA, in slots 12..23 (used 12 slots later)", d);
    run<0, 24, 0, false>("24 VALU / 24 MFMA", d);
    run<0, 48, 0, false>("48 VALU / 24 MFMA", d);
    run<0, 96, 0, false>("96 VALU / 24 MFMA", d);
    run<12, 48, 0, false>("12 ds_read_b128 + 48 VALU / 24 MFMA", d);
    return 0;
}
