// 256-point complex radix-4 DIF FFT building blocks (float64, LDS-resident), shared by the
// SIDEKIT and VBx front ends.  A 512-point real FFT is one 256-point complex FFT of
// z[n] = x[2n] + i x[2n+1] followed by the real-input untangle (see untangle_bin).
#pragma once
#include <hip/hip_runtime.h>

typedef double2 cplx;

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ cplx mul_mi(cplx a) { return make_double2(a.y, -a.x); }   // a * (-i)

// reverse the four base-4 digits of k (0..255): where the DIF output bin k lives
__device__ __forceinline__ int rev4(int k) {
    return ((k & 3) << 6) | (((k >> 2) & 3) << 4) | (((k >> 4) & 3) << 2) | ((k >> 6) & 3);
}

// first stage (span 256) from four register values a[m] = z[lane + 64 m]
__device__ __forceinline__ void fft256_stage0(cplx* z, int lane, const cplx a[4], const cplx* w256) {
    cplx s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]), s13 = cadd(a[1], a[3]), d13 = mul_mi(csub(a[1], a[3]));
    z[lane] = cadd(s02, s13);
    z[lane + 64] = cmul(cadd(d02, d13), w256[lane]);
    z[lane + 128] = cmul(csub(s02, s13), w256[(2 * lane) & 255]);
    z[lane + 192] = cmul(csub(d02, d13), w256[(3 * lane) & 255]);
}

// in-place radix-4 DIF butterfly on four points q apart, twiddle step `ts` (index into W256)
__device__ __forceinline__ void bfly4(cplx* z, int base, int q, int j, int ts, const cplx* w256) {
    cplx a0 = z[base + j], a1 = z[base + j + q], a2 = z[base + j + 2 * q], a3 = z[base + j + 3 * q];
    cplx s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    cplx y0 = cadd(s02, s13), y2 = csub(s02, s13), y1 = cadd(d02, d13), y3 = csub(d02, d13);
    if (ts > 0) {
        y1 = cmul(y1, w256[(ts * j) & 255]);
        y2 = cmul(y2, w256[(2 * ts * j) & 255]);
        y3 = cmul(y3, w256[(3 * ts * j) & 255]);
    }
    z[base + j] = y0; z[base + j + q] = y1; z[base + j + 2 * q] = y2; z[base + j + 3 * q] = y3;
}

// |X[k]|^2 of the 512-point real FFT from the digit-reversed 256-point complex result
__device__ __forceinline__ double untangle_power(const cplx* z, int k, const cplx* w512) {
    cplx zk = z[rev4(k)];
    cplx zm = z[rev4((256 - k) & 255)];
    zm.y = -zm.y;
    cplx e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y + zm.y));
    cplx d = csub(zk, zm);
    cplx o = make_double2(0.5 * d.y, -0.5 * d.x);          // -i/2 * (zk - zm)
    cplx x = cadd(e, cmul(w512[k], o));
    return __dadd_rn(__dmul_rn(x.x, x.x), __dmul_rn(x.y, x.y));
}

// host: fill W256 (256 complex) then W512 (256 complex), interleaved re/im doubles
static inline void fft256_host_twiddles(double* tw /* 1024 doubles */) {
    for (int k = 0; k < 256; ++k) {
        tw[2 * k] = cos(-2.0 * M_PI * k / 256.0);       tw[2 * k + 1] = sin(-2.0 * M_PI * k / 256.0);
        tw[512 + 2 * k] = cos(-2.0 * M_PI * k / 512.0); tw[512 + 2 * k + 1] = sin(-2.0 * M_PI * k / 512.0);
    }
}
