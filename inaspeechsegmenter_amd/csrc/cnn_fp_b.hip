// Instantiation unit of conv_x3_fp_kernel (conv_fp.h) for a subset of the filter shapes.
#include "conv_fp.h"

ISS_FP_DEFINE(5, 3)
