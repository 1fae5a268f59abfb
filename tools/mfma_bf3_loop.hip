// Micro-benchmark for the bf16x3 convolution row loop (round 2): the MFMA stream of one chunk of 8 channels of a
// k=5 Conv3d tile, fed from LDS exactly as planned for conv_bf3_kernel:
//   patch  [pos] 16 B hi plane + [pos] 16 B lo plane (8 channels as bf16), positions of a 4x8x16 tile + halo
//   weights [zd][g][kk][n][8 ch] hi and lo, g = group of 4 (zh,zw) taps (25 taps -> 7 groups)
//   K = 32 of v_mfma_f32_16x16x32_bf16 = 4 taps (lane group kk) x 8 channels; 3 MFMAs per product
//   (hi*hi + hi*lo + lo*hi), A_hi / A_lo / B_hi / B_lo are ds_read_b128.
// Prints the fp32-equivalent rate (one product = 2*16*16*32 FLOP per MFMA triple).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int PH = 12, PW = 20, PD = 8, NPOS = PD * PH * PW;      // 1920 positions
constexpr int NG = 7, KD = 5;

template <int MS, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void loop(float* out, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  bf16x8* Ahi = reinterpret_cast<bf16x8*>(lds);
  bf16x8* Alo = Ahi + NPOS;
  bf16x8* Bhi = Alo + NPOS;                       // [KD][NG][64 lanes]
  bf16x8* Blo = Bhi + KD * NG * 64;
  for (int i = threadIdx.x; i < 2 * NPOS + 2 * KD * NG * 64; i += THREADS) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(float)((i + j) & 3);
    Ahi[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  int pa[MS], toff[NG];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) {
    const int s = wave * MS + ms;                  // 32 sub-tiles of 16 W positions: (sd, sh)
    const int sd = s / 8, sh = s % 8;
    pa[ms] = (sd * PH + sh) * PW + i16;
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int t = min(4 * g + kk, 24);
    toff[g] = (t / 5) * PW + (t % 5);
  }
  f32x4 acc[MS];
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) acc[ms] = (f32x4){0, 0, 0, 0};
  for (int c = 0; c < chunks; ++c) {
    for (int zd = 0; zd < KD; ++zd) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const bf16x8 bh = Bhi[(zd * NG + g) * 64 + lane], bl = Blo[(zd * NG + g) * 64 + lane];
        bf16x8 ah[MS], al[MS];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          const int o = pa[ms] + toff[g] + zd * PH * PW;
          ah[ms] = Ahi[o]; al[ms] = Alo[o];
        }
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
          acc[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ms], bh, acc[ms], 0, 0, 0);
          acc[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ms], bl, acc[ms], 0, 0, 0);
          acc[ms] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ms], bh, acc[ms], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  float s = 0;
#pragma unroll
  for (int ms = 0; ms < MS; ++ms) s += acc[ms][0] + acc[ms][3];
  out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, double triples) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); for (int r = 0; r < 5; ++r) launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flop = triples * 2.0 * 16 * 16 * 32;
  printf("%-44s %.3f ms  %.0f TFLOP/s fp32-equivalent  (bf16 MFMA rate %.0f TFLOP/s = %.0f%% of 2500)\n", name, ms,
         flop / ms / 1e9, 3 * flop / ms / 1e9, 3 * flop / ms / 1e9 / 25.0);
}

int main() {
  float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
  const int chunks = 64, blocks = 1024;
  const size_t ldsb = (size_t)(2 * NPOS + 2 * KD * NG * 64) * 16;
  printf("LDS per block: %zu bytes\n", ldsb);
  {
    auto k = loop<8, 256>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    run("256 threads, MSUB 8 (1 wave / SIMD)", [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), ldsb, 0, out, chunks); },
        (double)blocks * chunks * KD * NG * 4 * 8);
  }
  {
    auto k = loop<4, 512>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    run("512 threads, MSUB 4 (2 waves / SIMD)", [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(512), ldsb, 0, out, chunks); },
        (double)blocks * chunks * KD * NG * 8 * 4);
  }
  {
    auto k = loop<2, 1024>;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    run("1024 threads, MSUB 2 (4 waves / SIMD)", [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), ldsb, 0, out, chunks); },
        (double)blocks * chunks * KD * NG * 16 * 2);
  }
  return 0;
}
