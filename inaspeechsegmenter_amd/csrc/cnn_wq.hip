// Instantiation unit of conv_x3_wq_kernel (conv_wq.h): one wave per SIMD, two LDS footprints; the fused 5x3 layer
// (f32 or CHL output; the fp16-operand forms are in cnn_wq_h.hip).
#include "conv_wq.h"

namespace issk {
void iss_wq_launch_5x3_f16(const ConvArgs& a, dim3 grid, hipStream_t st);
void iss_wq_launch_5x3(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if (a.f16) return iss_wq_launch_5x3_f16(a, grid, st);
    if (a.out_hl) hipLaunchKernelGGL((conv_x3_wq_kernel<5, 3, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_wq_kernel<5, 3, false>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
