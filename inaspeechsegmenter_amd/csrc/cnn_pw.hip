// Instantiation unit of conv_x3_pws_kernel (conv_pw.h): streaming pointwise GEMM, four epilogue forms.
#include "conv_pw.h"

namespace issk {
void iss_pws_launch(const ConvArgs& a, dim3 grid, hipStream_t st) {
    const bool simple = a.act <= 1 && !a.ps;
    if (a.res && simple) hipLaunchKernelGGL((conv_x3_pws_kernel<true, true>), grid, dim3(256), 0, st, a);
    else if (simple) hipLaunchKernelGGL((conv_x3_pws_kernel<false, true>), grid, dim3(256), 0, st, a);
    else if (a.res) hipLaunchKernelGGL((conv_x3_pws_kernel<true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_pws_kernel<false, false>), grid, dim3(256), 0, st, a);
}
void iss_pws2_launch(const ConvArgs& a0, hipStream_t st, bool strided, bool dual) {
    ConvArgs a = a0;
    a.nblk_n = (unsigned)(a.Cout / 128);
    const dim3 grid(std::min<unsigned>(a.nblk * a.nblk_n, 512u));
    if (dual) hipLaunchKernelGGL((conv_x3_pws2_kernel<true, false, true>), grid, dim3(256), 0, st, a);
    else if (strided) hipLaunchKernelGGL((conv_x3_pws2_kernel<true, true>), grid, dim3(256), 0, st, a);
    else if (a.act <= 1 && !a.ps) hipLaunchKernelGGL((conv_x3_pws2_kernel<true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_pws2_kernel<false>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
