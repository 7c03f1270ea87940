// Instantiation unit of conv_x3_pwc_kernel (conv_pwc.h): chained expansion + reduction of two consecutive Bottlenecks.
#include "conv_pwc.h"

namespace issk {
void iss_pwc_launch(const ConvArgs& a, hipStream_t st) {
    const dim3 grid(std::min<unsigned>(a.nblk, 256u));       // one workgroup per CU (155 KB of LDS)
    // C1 = 32: 68 / 73 KB of LDS and <= 252 registers -- TWO workgroups fit a CU, and the second one's waves run their epilogue / split /
    // staging phases under the first one's MFMAs (profiles/r06_pwc_experiments.txt: those phases, not memory, bind the kernel)
    const dim3 grid2(std::min<unsigned>(a.nblk, 512u));
    if (a.Cin == 32 && a.Cout2 == 32) hipLaunchKernelGGL((conv_x3_pwc_kernel<1, 1>), grid2, dim3(256), 0, st, a);
    else if (a.Cin == 32) hipLaunchKernelGGL((conv_x3_pwc_kernel<1, 2>), grid2, dim3(256), 0, st, a);
    else if (a.Cin == 64 && a.Cout2 == 64) hipLaunchKernelGGL((conv_x3_pwc_kernel<2, 2>), grid, dim3(256), 0, st, a);
    else if (a.Cin == 64) hipLaunchKernelGGL((conv_x3_pwc_kernel<2, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_x3_pwc_kernel<4, 4>), grid, dim3(256), 0, st, a);
}
}  // namespace issk
