// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the ring form for 5x5 second convolutions (25 taps), first-layer-fused.
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_ring_5x5(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) { launch_ws_fused_rowmajor<5, 5, false>(a, grid, st, padded); }
}  // namespace issk
