// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the plain (unfused) zero-padded 3x3 variant.
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_plain_3x3(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if (epi_is_simple_tr(a)) hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, true, false, 1, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_ws_kernel<3, 3, true, true, false>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
