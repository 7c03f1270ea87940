// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): 3x3 behind a zero-padded ('same') shared first layer (FS).
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_fs_3x3(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) { launch_ws_fused_rowmajor<3, 3, true>(a, grid, st, padded); }
}  // namespace issk
