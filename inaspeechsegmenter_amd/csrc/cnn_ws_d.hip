// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the two-column-half (NH = 2) variant for unpadded 3x3 layers.
#include "conv_ws.h"

namespace issk {
// zero-padded form (ResNet-101's 128 -> 128 / 256 -> 256 convolutions): transposed, simple epilogue only (host-checked)
void iss_ws_launch_nh2_3x3_padded(const ConvArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, true, false, 2, 1>), grid, dim3(512), 0, st, a);
}
// zero-padded + relu + fused non-overlapping max-pool (a 'same' 3x3 layer in front of a pool: VGG-style stacks), row-major epilogue
void iss_ws_launch_nh2_3x3_padded_pool(const ConvArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, false, false, 2, 1>), grid, dim3(512), 0, st, a);
}
void iss_ws_launch_nh2_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr) {
    if (tr && epi_is_simple_tr(a)) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, true, false, 2, 1>), grid, dim3(512), 0, st, a);
    else if (tr) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, true, false, 2>), grid, dim3(512), 0, st, a);
    else if (epi_is_pool_relu(a)) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, false, false, 2, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, false, false, 2>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
