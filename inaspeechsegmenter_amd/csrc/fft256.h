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

// LDS slot of logical point i of the 256-point work array.  One cplx is 16 bytes; a ds_read_b128 is serviced in four groups
// of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32: MI355X_MICROARCH.md), a ds_write_b128 in eight groups of 8
// contiguous lanes, and 16 consecutive points cover all 64 banks once: a group is conflict-free iff its points fall into
// different (slot mod 16).  The identity layout serves only the first two stages: the span-16 and span-4 stages are 4-way
// and the digit-reversed untangle reads up to 16-way conflicted (rocprofv3 counted 49 % of the LDS cycles of the front-end
// kernels as bank conflicts).  Slot = i with its low nibble XOR-ed by a GF(2)-linear function of its high nibble; the
// matrix below is the best of all 65 536 such maps under the real lane groups (exhaustive search over every access of a
// frame: 224 LDS-array cycles against 600 for the identity and 208 for a conflict-free layout).
__device__ __forceinline__ int zslot(int i) {
    return i ^ ((i & 16) ? 9 : 0) ^ ((i & 32) ? 12 : 0) ^ ((i & 64) ? 2 : 0) ^ ((i & 128) ? 1 : 0);
}

// first stage (span 256) from four register values a[m] = z[lane + 64 m]
__device__ __forceinline__ void fft256_stage0(cplx* z, int lane, const cplx a[4], const cplx* w256) {
    cplx s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]), s13 = cadd(a[1], a[3]), d13 = mul_mi(csub(a[1], a[3]));
    z[zslot(lane)] = cadd(s02, s13);
    z[zslot(lane + 64)] = cmul(cadd(d02, d13), w256[lane]);
    z[zslot(lane + 128)] = cmul(csub(s02, s13), w256[(2 * lane) & 255]);
    z[zslot(lane + 192)] = cmul(csub(d02, d13), w256[(3 * lane) & 255]);
}

// twiddles of one lane's butterfly in a later stage: they depend on the lane only, so they live in registers across frames
struct Tw3 { cplx w1, w2, w3; };
__device__ __forceinline__ Tw3 bfly_twiddles(int j, int ts, const cplx* w256) {
    Tw3 t;
    t.w1 = w256[(ts * j) & 255]; t.w2 = w256[(2 * ts * j) & 255]; t.w3 = w256[(3 * ts * j) & 255];
    return t;
}

// in-place radix-4 DIF butterfly on four points q apart; TW: multiply outputs 1..3 by the lane's twiddles
template <bool TW>
__device__ __forceinline__ void bfly4(cplx* z, int base, int q, int j, const Tw3& tw) {
    const int i0 = zslot(base + j), i1 = zslot(base + j + q), i2 = zslot(base + j + 2 * q), i3 = zslot(base + j + 3 * q);
    cplx a0 = z[i0], a1 = z[i1], a2 = z[i2], a3 = z[i3];
    cplx s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = mul_mi(csub(a1, a3));
    cplx y0 = cadd(s02, s13), y2 = csub(s02, s13), y1 = cadd(d02, d13), y3 = csub(d02, d13);
    if (TW) { y1 = cmul(y1, tw.w1); y2 = cmul(y2, tw.w2); y3 = cmul(y3, tw.w3); }
    z[i0] = y0; z[i1] = y1; z[i2] = y2; z[i3] = y3;
}

// |X[k]|^2 of the 512-point real FFT from the digit-reversed 256-point complex result
__device__ __forceinline__ double untangle_power(const cplx* z, int k, const cplx* w512) {
    cplx zk = z[zslot(rev4(k))];
    cplx zm = z[zslot(rev4((256 - k) & 255))];
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
