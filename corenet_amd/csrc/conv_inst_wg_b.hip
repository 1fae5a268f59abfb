// Explicit instantiations of the conv engine kernels (split for parallel compilation).
#include "conv_kernels.h"

int crn_launch_wgrad_4_2(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<4, 2>(g, grid, lds, st); }
int crn_launch_wgrad_4_1(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<4, 1>(g, grid, lds, st); }
int crn_launch_wgrad_2_4(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<2, 4>(g, grid, lds, st); }
