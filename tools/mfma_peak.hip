// Micro-benchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, int EXTRA>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float x = a + threadIdx.x, y = b;
  int z = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < EXTRA; ++e) asm volatile("v_add_u32 %0, %0, 1" : "+v"(z));
      }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + z;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop_per_block_iter, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-28s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flop_per_block_iter * blocks * iters / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int blocks : {256, 512, 1024}) {
    run("16x16x4 acc8", [&] { hipLaunchKernelGGL((k16<8, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 8 * 2048, blocks, iters);
    run("16x16x4 acc4", [&] { hipLaunchKernelGGL((k16<4, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 4 * 2048, blocks, iters);
    run("16x16x4 acc8 +1valu", [&] { hipLaunchKernelGGL((k16<8, 1>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 8 * 2048, blocks, iters);
    run("16x16x4 acc8 +3valu", [&] { hipLaunchKernelGGL((k16<8, 3>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 8 * 2048, blocks, iters);
    run("16x16x4 acc8 +6valu", [&] { hipLaunchKernelGGL((k16<8, 6>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 8 * 2048, blocks, iters);
    run("32x32x2 acc4", [&] { hipLaunchKernelGGL((k32<4>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); }, 4.0 * 4 * 4 * 4096, blocks, iters);
  }
  return 0;
}
