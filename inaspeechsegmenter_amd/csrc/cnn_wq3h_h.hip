// conv_x3_wq3h_kernel with fp16 operand halves (ISS_PREC_F16X3, conv_common.h).
#include "conv_wq3h.h"

namespace issk {
void iss_wq3h_launch_f16(const ConvArgs& a, dim3 grid, hipStream_t st, int kind) {
    if (kind == 0 && a.out_hl) hipLaunchKernelGGL((conv_x3_wq3h_kernel<0, true, true>), grid, dim3(256), 0, st, a);
    else if (kind == 0) hipLaunchKernelGGL((conv_x3_wq3h_kernel<0, false, true>), grid, dim3(256), 0, st, a);
    else if (a.out_hl) hipLaunchKernelGGL((conv_x3_wq3h_kernel<1, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_wq3h_kernel<1, false, true>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
