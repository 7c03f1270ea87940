// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the exact-f32 form (F32, v_mfma_f32_32x32x2_f32) for the three layers that carry
// the segmenter nets' arithmetic -- the first-layer-fused 5x3 convolution and the two unpadded 3x3 layers (128 columns per workgroup).
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_f32_fused_5x3(const ConvArgs& a, dim3 grid, hipStream_t st) {      // relu + 2x2 max-pool (epi_is_pool_relu)
    hipLaunchKernelGGL((conv_x3_ws_kernel<5, 3, false, false, true, 1, 1, false, true>), grid, dim3(512), 0, st, a);
}
void iss_ws_launch_f32_nh2_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr) {  // tr: bias + relu, transposed; else pooled relu
    if (tr) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, true, false, 2, 1, false, true>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, false, false, 2, 1, false, true>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
