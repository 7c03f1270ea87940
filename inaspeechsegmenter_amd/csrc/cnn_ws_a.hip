// Instantiation unit of conv_x3_ws_kernel (conv_ws.h): weight-stationary footprint kernel, filter shapes with 8..16 taps.
#include "conv_ws.h"

#define ISS_WS_DEFINE(KH_, KW_)                                                                                          \
    void iss_ws_launch_##KH_##x##KW_(const issk::ConvArgs& a, dim3 grid, hipStream_t st, bool padded, bool tr, bool fused) { \
        issk::launch_ws_shape<KH_, KW_>(a, grid, st, padded, tr, fused);                                                  \
    }
ISS_WS_SHAPES_A(ISS_WS_DEFINE)
