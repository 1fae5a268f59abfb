// Explicit instantiations of the conv engine kernels (split for parallel compilation).
#include "conv_kernels.h"

int crn_launch_fwd_8_1(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<8, 1>(g, grid, lds, st); }
