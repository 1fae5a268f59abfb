// Explicit instantiations of the conv engine kernels (split for parallel compilation).
#include "conv_kernels.h"

int crn_launch_wgrad_2_2(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<2, 2>(g, grid, lds, st); }
int crn_launch_wgrad_2_1(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<2, 1>(g, grid, lds, st); }
int crn_launch_wgrad_1_4(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<1, 4>(g, grid, lds, st); }
int crn_launch_wgrad_1_2(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<1, 2>(g, grid, lds, st); }
int crn_launch_wgrad_1_1(const crnk::WgradGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_wgrad<1, 1>(g, grid, lds, st); }
