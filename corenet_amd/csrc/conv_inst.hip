// Explicit instantiation of one conv-engine tile configuration (kernel templates: conv_kernels.h).
// build.py compiles this file once per configuration, in parallel:
//   -DCRN_INST_FWD   -DCRN_M=<MSUB> -DCRN_N=<NSUB>   forward / data-grad kernels (scalar + 16-byte staging)
//   -DCRN_INST_WGRAD -DCRN_M=<RSUB> -DCRN_N=<NSUB>   weight-grad kernels (scalar, x 16-byte, x+dy 16-byte, x 16-byte + dy pairs)
#include "conv_kernels.h"

#define CRN_CAT3(a, b, c) a##b##_##c
#define CRN_NAME(prefix, M, N) CRN_CAT3(prefix, M, N)

#if defined(CRN_INST_FWD)
int CRN_NAME(crn_launch_fwd_, CRN_M, CRN_N)(const crnk::ConvGeom& g, int xvec, dim3 grid, size_t lds, hipStream_t st) {
  // xvec: 0 scalar staging, 1 float4 units (unit-stride x), 2 position pairs (stride-2 space-to-depth x)
  if (xvec == 1) return crnk::launch_fwd<CRN_M, CRN_N, 1>(g, grid, lds, st);
  if (xvec == 2) return crnk::launch_fwd<CRN_M, CRN_N, 2>(g, grid, lds, st);
  return crnk::launch_fwd<CRN_M, CRN_N, 0>(g, grid, lds, st);
}
#elif defined(CRN_INST_WGRAD)
int CRN_NAME(crn_launch_wgrad_, CRN_M, CRN_N)(const crnk::WgradGeom& g, int xvec, int dyvec, dim3 grid, size_t lds,
                                              hipStream_t st) {
  // dyvec: 0 scalar, 1 float4 units (unit-stride dy), 2 position pairs (stride-2 space-to-depth dy)
  if (xvec && dyvec == 1) return crnk::launch_wgrad<CRN_M, CRN_N, true, 1>(g, grid, lds, st);
  if (xvec && dyvec == 2) return crnk::launch_wgrad<CRN_M, CRN_N, true, 2>(g, grid, lds, st);
  if (xvec) return crnk::launch_wgrad<CRN_M, CRN_N, true, 0>(g, grid, lds, st);
  return crnk::launch_wgrad<CRN_M, CRN_N, false, 0>(g, grid, lds, st);
}
#else
#error "define CRN_INST_FWD or CRN_INST_WGRAD"
#endif
