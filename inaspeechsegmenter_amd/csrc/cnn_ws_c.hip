// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the plain (unfused) zero-padded 3x3 variant.
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_plain_3x3(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if (epi_is_simple_tr(a)) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, true, false, 1, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, true, false>), grid, dim3(512), 0, st, a);
}
// unpadded, not first-layer-fused, one 64-column tile: 3x3 layers with 64 / 96 output channels (the two-column-half form wants % 128)
void iss_ws_launch_plain_3x3_unpadded(const ConvArgs& a, dim3 grid, hipStream_t st, bool tr) {
    if (tr) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, true, false, 1, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, false, false, false, 1, 1>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
