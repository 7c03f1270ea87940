// Does v_mfma_f32_32x32x16_f16 honour SUBNORMAL f16 inputs on gfx950?  (decides whether an f16 hi/lo operand split keeps its lo
// parts: for |x| < 0.125 the lo part of x = hi + lo is below f16's smallest normal 6.1e-5).   hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* av, const float* bv, float* out) {
    const int lane = threadIdx.x;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row = lane & 31][k = 8 * (lane >> 5) + i], B[k][col = lane & 31]: put one value at k = 0 of every row / column
    if (lane < 32) { a[0] = (_Float16)av[0]; b[0] = (_Float16)bv[0]; }
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
}
int main() {
    float *da, *db, *dout;
    hipMalloc(&da, 4); hipMalloc(&db, 4); hipMalloc(&dout, 4);
    const float as[] = {1.0f, 3.0e-5f, 1.0e-6f, 6.0e-8f, 3.0e-5f, 2.0e-7f};
    const float bs[] = {1.0f, 1.0f,    1.0f,    1.0f,    1024.f,  4096.f};
    for (int t = 0; t < 6; ++t) {
        hipMemcpy(da, &as[t], 4, hipMemcpyHostToDevice); hipMemcpy(db, &bs[t], 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
        float o; hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost);
        const float want = (float)(_Float16)as[t] * (float)(_Float16)bs[t];
        printf("a = %.3e (f16 %.6e) x b = %.1f -> mfma %.6e  want %.6e  %s\n", as[t], (double)(float)(_Float16)as[t], bs[t], o, want,
               o == want ? "exact" : (o == 0.f ? "FLUSHED" : "differs"));
    }
    return 0;
}
