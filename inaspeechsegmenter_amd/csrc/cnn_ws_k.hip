// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): NCB = 1 -- one 32-column block per workgroup -- for a first-layer-fused 5x3
// second convolution with <= 32 output channels (relu + 2x2 max-pool).
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_ncb1_5x3(const ConvArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_x3_ws_kernel<5, 3, false, false, true, 1, 1, false, false, 1>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
