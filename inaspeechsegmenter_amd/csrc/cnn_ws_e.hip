// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): the ring form for filters with more than 16 taps (7x7), first-layer-fused.
#include "conv_ws.h"

namespace issk {
void iss_ws_launch_ring_7x7(const ConvArgs& a, dim3 grid, hipStream_t st, bool padded) { launch_ws_fused_rowmajor<7, 7, false>(a, grid, st, padded); }
}  // namespace issk
