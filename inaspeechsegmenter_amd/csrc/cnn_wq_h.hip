// conv_x3_wq_kernel with fp16 operand halves (ISS_PREC_F16X3, conv_common.h): its own unit so that `make -j` builds it beside cnn_wq.hip.
#include "conv_wq.h"

namespace issk {
void iss_wq_launch_5x3_f16(const ConvArgs& a, dim3 grid, hipStream_t st) {
    if (a.out_hl) hipLaunchKernelGGL((conv_x3_wq_kernel<5, 3, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_wq_kernel<5, 3, false, true>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
