// Micro-benchmark for the round-2 plan (bf16x3 split MFMA): sustained rate of the bf16 MFMA instructions of
// gfx950, bare and fed from LDS (8-byte operand reads), and the resulting fp32-equivalent rate when every
// product needs 3 MFMAs (hi*hi + hi*lo + lo*hi).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256, 2) void bare16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  s16x4 a = {(short)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (short)blockIdx.x};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void bare32(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(blockIdx.x + j); }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// LDS-fed: per "tap" MS A-reads + 1 B-read of 16 bytes (8 bf16) each, MS MFMAs 16x16x32
template <int MS>
__global__ __launch_bounds__(256, 2) void lds32(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (__bf16)(float)(i & 7);
  __syncthreads();
  f32x4 acc[MS];
  for (int m = 0; m < MS; ++m) acc[m] = (f32x4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const bf16x8* pa = reinterpret_cast<const bf16x8*>(lds) + (lane & 15) + (lane >> 4) * 40;
  const bf16x8* pb = reinterpret_cast<const bf16x8*>(lds + 8192) + (lane & 15) + (lane >> 4) * 36;
  for (int it = 0; it < iters; ++it) {
    const int ro = (it & 7) * 18;
    bf16x8 a[5][MS], b[5];
#pragma unroll
    for (int z = 0; z < 5; ++z) {
#pragma unroll
      for (int m = 0; m < MS; ++m) a[z][m] = pa[ro + m * 64 + z];
      b[z] = pb[ro + z * 16];
    }
#pragma unroll
    for (int z = 0; z < 5; ++z)
#pragma unroll
      for (int m = 0; m < MS; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[z][m], b[z], acc[m], 0, 0, 0);
  }
  float s = 0; for (int m = 0; m < MS; ++m) s += acc[m][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flop) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-40s %.3f ms  %.0f TFLOP/s bf16  (%.0f TFLOP/s fp32-equivalent at 3 MFMAs per product)\n", name, ms,
         flop / ms / 1e9, flop / ms / 1e9 / 3);
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000, blocks = 512;
  const double w = 4.0 * blocks * iters;
  run("bare 16x16x16 bf16_1k, 8 acc", [&] { hipLaunchKernelGGL((bare16<8>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 8 * 8192);
  run("bare 16x16x32 bf16, 8 acc", [&] { hipLaunchKernelGGL((bare32<8>), dim3(blocks), dim3(256), 0, 0, out, iters); }, w * 8 * 16384);
  run("LDS-fed 16x16x32 bf16 MS8 (5 taps)", [&] { hipLaunchKernelGGL((lds32<8>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, w / 4 * 5 * 8 * 16384);
  run("LDS-fed 16x16x32 bf16 MS4 (5 taps)", [&] { hipLaunchKernelGGL((lds32<4>), dim3(blocks), dim3(256), 0, 0, out, iters / 4); }, w / 4 * 5 * 4 * 16384);
  return 0;
}
