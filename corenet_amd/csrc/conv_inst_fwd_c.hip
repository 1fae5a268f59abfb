// Explicit instantiations of the conv engine kernels (split for parallel compilation).
#include "conv_kernels.h"

int crn_launch_fwd_2_2(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<2, 2>(g, grid, lds, st); }
int crn_launch_fwd_2_1(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<2, 1>(g, grid, lds, st); }
int crn_launch_fwd_1_4(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<1, 4>(g, grid, lds, st); }
int crn_launch_fwd_1_2(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<1, 2>(g, grid, lds, st); }
int crn_launch_fwd_1_1(const crnk::ConvGeom& g, dim3 grid, size_t lds, hipStream_t st) { return crnk::launch_fwd<1, 1>(g, grid, lds, st); }
