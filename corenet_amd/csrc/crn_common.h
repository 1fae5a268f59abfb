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

// Block-wide sums of two doubles per thread with one barrier pair (thread 0 gets both); `smem`: 2 * blockDim.x/64.
__device__ __forceinline__ void crn_block_sum2(double a, double b, double* smem, double& ra, double& rb) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  a = crn_wave_sum(a);
  b = crn_wave_sum(b);
  __syncthreads();
  if (lane == 0) { smem[wid] = a; smem[nw + wid] = b; }
  __syncthreads();
  ra = 0.0; rb = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) { ra += smem[i]; rb += smem[nw + i]; }
}

// Kernel arguments live in host-visible memory: a scalar load of one costs 0.2-0.35 us (measured with shader-clock
// stamps in conv_e2d.hip), and the compiler places each s_load next to the first use of the argument -- in kernels
// with control flow that becomes 4-8 dependent round trips before the first global load is issued.  Naming the
// arguments in the entry block turns them into one.
template <typename A>
__device__ __forceinline__ int crn_kernarg_now(const A& a) { asm volatile("" ::"s"(a)); return 0; }
template <typename... A>
__device__ __forceinline__ void crn_kernargs_now(const A&... a) { const int u[] = {crn_kernarg_now(a)...}; (void)u; }
// The same for a by-value argument struct too large to keep in SGPRs: one dword per 64-byte line of it is loaded
// in the entry block, which pulls the whole struct into the scalar cache in ONE round trip; the loads the compiler
// places later (and the re-loads it emits under SGPR pressure) then hit the cache.
template <typename T>
__device__ __forceinline__ void crn_kernarg_touch(const T& g) {
  const unsigned* p = reinterpret_cast<const unsigned*>(&g);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i += 16) asm volatile("" ::"s"(p[i]));
  asm volatile("" ::"s"(p[sizeof(T) / 4 - 1]));
}

// Deterministic mode (env CRN_DETERMINISTIC=1 or crn_set_deterministic(1)): every floating-point sum of the library is
// taken in an order that does not depend on how workgroups are scheduled, so two runs from the same state are
// bit-identical (debugging aid: a loss regression can be told from atomic-order noise).  Weight gradients give each
// dw element to ONE workgroup (no split over the positions: slow), the bias gradient fused into the BatchRenorm
// backward becomes its own ordered reduction, and the ray-sample scatter accumulates in 64-bit fixed point.
bool crn_deterministic();

// A split-K convolution whose partial sums are still in the scratch: the next BatchRenorm launch over `y` adds them up
// itself (crn_splitk_defer); anything else that needs the scratch or y first calls crn_splitk_flush.
struct CrnSplitPending {
  bool armed = false, active = false;
  crnView y{};                 // the conv's real output (dense [B][C][S])
  const float* scratch = nullptr;
  int splits = 0;
  hipStream_t stream = nullptr;  // the stream (and its device) the convolution ran on: only a BatchRenorm call on the SAME
  int device = -1;               // stream may take the sum over; a flush always runs on this stream, behind the conv
};
CrnSplitPending& crn_splitk_pending();                 // per host thread
// Run the pending reduction now, on the stream of the convolution that left it (no-op if none).  EVERY entry point of
// the library calls this first (CRN_ENTRY(stream)), except the BatchRenorm calls that can take the sum over.
int crn_splitk_flush(hipStream_t st);
void crn_splitk_set_pending(const crnView& y, const float* scratch, int splits, hipStream_t st);
#define CRN_ENTRY(stream_arg)                                                          \
  do {                                                                                 \
    const int crn_rcf_ = crn_splitk_flush((hipStream_t)(stream_arg));                  \
    if (crn_rcf_ != CRN_OK) return crn_rcf_;                                           \
  } while (0)
bool crn_splitk_take_armed();                          // consumes the arming of crn_splitk_defer
int crn_splitk_reduce_view(const crnView& y, const float* scratch, int splits, int accumulate, hipStream_t st);
