// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): 5x3 behind a zero-padded ('same') shared first layer (FS).
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_fs_5x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) { launch_ws_fused_rowmajor<5, 3, true>(a, grid, st, padded); }
void iss_ws_launch_fs_5x3_tr(const ConvArgs& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_x3_ws_kernel<5, 3, false, true, true, 1, 1, true>), grid, dim3(512), 0, st, a);
}
}  // namespace issk
