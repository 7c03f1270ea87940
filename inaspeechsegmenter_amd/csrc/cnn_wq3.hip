// Instantiation unit of conv_x3_wq3_kernel (conv_wq3.h): one wave per SIMD, the unpadded 3x3 layers with 128 output channels
// per workgroup; kind 0 = bias + relu (transposed accumulators), kind 1 = relu + 2 x 1 max-pool.
#include "conv_wq3.h"

namespace issk {
void iss_wq3_launch(const ConvArgs& a, dim3 grid, hipStream_t st, int kind) {
    if (kind == 0) hipLaunchKernelGGL((conv_x3_wq3_kernel<0>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_wq3_kernel<1>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
