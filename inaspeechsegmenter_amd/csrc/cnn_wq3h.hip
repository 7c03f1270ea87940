// Instantiation unit of conv_x3_wq3h_kernel (conv_wq3h.h): the one-wave-per-SIMD 3x3 kernel on a CHL (pre-split, plane-major)
// input fetched by LDS-DMA; kind 0 = bias + relu with f32 or CHL output, kind 1 = relu + 2 x 1 max-pool with f32 output.
// (The fp16-operand forms are in cnn_wq3h_h.hip.)
#include "conv_wq3h.h"

namespace issk {
void iss_wq3h_launch_f16(const ConvArgs& a, dim3 grid, hipStream_t st, int kind);
void iss_wq3h_launch(const ConvArgs& a, dim3 grid, hipStream_t st, int kind) {
    if (a.f16) return iss_wq3h_launch_f16(a, grid, st, kind);
    if (kind == 0 && a.out_hl) hipLaunchKernelGGL((conv_x3_wq3h_kernel<0, true>), grid, dim3(256), 0, st, a);
    else if (kind == 0) hipLaunchKernelGGL((conv_x3_wq3h_kernel<0, false>), grid, dim3(256), 0, st, a);
    else if (a.out_hl) hipLaunchKernelGGL((conv_x3_wq3h_kernel<1, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_wq3h_kernel<1, false>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
