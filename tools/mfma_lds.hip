// Micro-benchmark: MFMA stream fed from LDS the way the conv row loop does it (MS A-reads + NS B-reads per tap,
// MS*NS MFMAs), 16x16x4 vs 32x32x2 tiles.  Upper bound for what a 32x32 variant of the conv kernels could gain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MS, int NS, int KW>
__global__ __launch_bounds__(256, 2) void k16(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x4 acc[MS][NS];
  for (int m = 0; m < MS; ++m) for (int n = 0; n < NS; ++n) acc[m][n] = (f32x4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const float* pa = lds + (lane & 15) + (lane >> 4) * 400;
  const float* pb = lds + 4096 + (lane & 15) + (lane >> 4) * 300;
  for (int it = 0; it < iters; ++it) {
    const int ro = (it & 15) * 20;
    float a[KW][MS], b[KW][NS];
#pragma unroll
    for (int z = 0; z < KW; ++z) {
#pragma unroll
      for (int m = 0; m < MS; ++m) a[z][m] = pa[ro + m * 64 + z];
#pragma unroll
      for (int n = 0; n < NS; ++n) b[z][n] = pb[ro + n * 16 + z * 32];
    }
#pragma unroll
    for (int z = 0; z < KW; ++z)
#pragma unroll
      for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[z][m], b[z][n], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MS + NS, 0);
#pragma unroll
    for (int z = 0; z < KW; ++z) { __builtin_amdgcn_sched_group_barrier(0x100, MS + NS, 0); __builtin_amdgcn_sched_group_barrier(0x008, MS * NS, 0); }
  }
  float s = 0;
  for (int m = 0; m < MS; ++m) for (int n = 0; n < NS; ++n) s += acc[m][n][0] + acc[m][n][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MS, int NS, int KW>
__global__ __launch_bounds__(256, 2) void k32(float* out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)(i & 7);
  __syncthreads();
  f32x16 acc[MS][NS];
  for (int m = 0; m < MS; ++m) for (int n = 0; n < NS; ++n) for (int j = 0; j < 16; ++j) acc[m][n][j] = 0;
  const int lane = threadIdx.x & 63;
  const float* pa = lds + (lane & 31) + (lane >> 5) * 400;
  const float* pb = lds + 4096 + (lane & 31) + (lane >> 5) * 300;
  for (int it = 0; it < iters; ++it) {
    const int ro = (it & 15) * 20;
    float a[KW][MS], b[KW][NS];
#pragma unroll
    for (int z = 0; z < KW; ++z) {
#pragma unroll
      for (int m = 0; m < MS; ++m) a[z][m] = pa[ro + m * 64 + z];
#pragma unroll
      for (int n = 0; n < NS; ++n) b[z][n] = pb[ro + n * 32 + z * 64];
    }
#pragma unroll
    for (int z = 0; z < KW; ++z)
#pragma unroll
      for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[z][m], b[z][n], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, MS + NS, 0);
#pragma unroll
    for (int z = 0; z < KW; ++z) { __builtin_amdgcn_sched_group_barrier(0x100, MS + NS, 0); __builtin_amdgcn_sched_group_barrier(0x008, MS * NS, 0); }
  }
  float s = 0;
  for (int m = 0; m < MS; ++m) for (int n = 0; n < NS; ++n) s += acc[m][n][0] + acc[m][n][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 2000, blocks = 512;
  const double w = 4.0 * blocks * iters;   // waves * iters
  run("16x16x4  MS8 NS1 KW5", [&] { hipLaunchKernelGGL((k16<8, 1, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 8 * 2048);
  run("16x16x4  MS4 NS2 KW5", [&] { hipLaunchKernelGGL((k16<4, 2, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 8 * 2048);
  run("16x16x4  MS4 NS1 KW5", [&] { hipLaunchKernelGGL((k16<4, 1, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 4 * 2048);
  run("32x32x2  MS2 NS1 KW5", [&] { hipLaunchKernelGGL((k32<2, 1, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 2 * 4096);
  run("32x32x2  MS4 NS1 KW5", [&] { hipLaunchKernelGGL((k32<4, 1, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 4 * 4096);
  run("32x32x2  MS2 NS2 KW5", [&] { hipLaunchKernelGGL((k32<2, 2, 5>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 5 * 4 * 4096);
  return 0;
}
