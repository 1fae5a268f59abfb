// Explicit instantiations of the conv engine kernels (split for parallel compilation).
#include "conv_kernels.h"

int crn_launch_wgrad_8_1(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<8, 1>(g, grid, lds, st); }
