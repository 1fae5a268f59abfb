// Shared device/host helpers for libcorenet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/corenet_hip.h"

#define CRN_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

#define CRN_HIP(expr)                            \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int crn_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// wave64 sum reduction (all lanes get the result of lane 0's tree; use lane 0)
__device__ __forceinline__ double crn_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float crn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Block-wide sum of one double per thread; result valid in thread 0.
// `smem` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double crn_block_sum(double v, double* smem) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  v = crn_wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[wid] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}
