// Instantiation unit of conv_dhl_kernel (conv_dhl.h): the first dense layer of the segmenter nets on the pre-split (PHL) tensor conv4
// writes for it; bf16 and fp16 operand halves.
#include "conv_dhl.h"

namespace issk {
// weights [Cout][Kpad] (16-bit hi and lo arrays) -> the packed order above; columns >= Cout are zero
__global__ void dhl_pack_kernel(const uint16_t* __restrict__ wh, const uint16_t* __restrict__ wl, uint16_t* __restrict__ out,
                                int Cout, int Kpad, int K) {
    const long long per_tile = (long long)(K / 8) * 2 * DHL_BN * 8;
    const long long total = per_tile * dhl_col_tiles(Cout);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ytile = (int)(i / per_tile);
        const long long j = i - (long long)ytile * per_tile;
        const int e = (int)(j & 7);
        long long t = j >> 3;
        const int n = ytile * DHL_BN + (int)(t % DHL_BN); t /= DHL_BN;
        const int part = (int)(t & 1);
        const int kg = (int)(t >> 1);                                   // k-group of 8 (k-tile * 4 + group)
        const int k = kg * 8 + e;
        out[i] = n < Cout ? (part ? wl : wh)[(size_t)n * Kpad + k] : (uint16_t)0;
    }
}

void iss_dhl_pack(const uint16_t* wh, const uint16_t* wl, uint16_t* out, int Cout, int Kpad, int K, hipStream_t st) {
    const long long total = (long long)dhl_packed_elems(K, Cout);
    hipLaunchKernelGGL(dhl_pack_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 4096)), dim3(256), 0, st, wh, wl, out, Cout, Kpad, K);
}
void iss_dhl_launch(const DhlArgs& a, hipStream_t st, bool f16) {
    const dim3 grid((unsigned)((a.M + DHL_BM - 1) / DHL_BM), (unsigned)dhl_col_tiles(a.Cout));
    if (f16) hipLaunchKernelGGL((conv_dhl_kernel<true>), grid, dim3(ISS_DHL_NW * 64), 0, st, a);
    else hipLaunchKernelGGL((conv_dhl_kernel<false>), grid, dim3(ISS_DHL_NW * 64), 0, st, a);
}
}  // namespace issk
