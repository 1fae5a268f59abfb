// LDS atomic throughput on gfx950, one number per (op, active lanes, address pattern): shader cycles per wave instruction with
// 4 / 8 / 16 waves per CU issuing the same stream.  Built and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_probe.hip -o tools/_build/lds_atomic_probe && tools/_build/lds_atomic_probe
// Why: the ray-sample scatter spends 35 of its 48 us in ds_add_f32 (profiles/r05_ray_sweep.txt, CRN_RAY_DBG=1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void probe(int iters, int stride_mask, int lane_step, float* sink, long long* cycles) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  // address pattern: stride_mask 0 -> unique address per lane; 1 -> pairs of lanes share; 3 -> quads share; 63 -> all lanes the same
  const int slot = (lane & ~stride_mask) + wave * 64;
  const bool active = (lane % lane_step) == 0;
  float v = (float)(lane + 1) * 0.25f;
  const long long t0 = __builtin_readcyclecounter();
  if (active) {
    for (int i = 0; i < iters; ++i) {
      const int a = (slot + (i & 15) * 256) & 8191;          // walk 16 windows so that consecutive ops are independent addresses
      if (OP == 0) { atomicAdd(&lds[a], v); }
      else if (OP == 1) { atomicAdd(reinterpret_cast<unsigned*>(&lds[a]), (unsigned)lane); }
      else if (OP == 2) { atomicAdd(reinterpret_cast<unsigned long long*>(&lds[(a * 2) & 8190]), (unsigned long long)lane); }
      else if (OP == 3) { float r = lds[a]; r += v; lds[a] = r; }
      else if (OP == 4) { float4 r = *reinterpret_cast<float4*>(&lds[(a * 4) & 8188]); r.x += v; r.y += v; r.z += v; r.w += v; *reinterpret_cast<float4*>(&lds[(a * 4) & 8188]) = r; }
      else if (OP == 5) { v += __shfl_down(v, 1); }          // ds_bpermute round trip
      else if (OP == 6) { asm volatile("ds_add_rtn_f32 %0, %1, %2\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a * 4), "v"(v) : "memory"); }
      else if (OP == 7) { asm volatile("ds_pk_add_f16 %0, %1" ::"v"(a * 4), "v"(__float_as_uint(v)) : "memory"); }
    }
  }
  __syncthreads();
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
  float s = v;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s += lds[i];
  if (s == 123.456f) sink[0] = s;
}

template <int OP>
void run(const char* name, float* sink, long long* cyc) {
  const int iters = 4096;
  for (int wg_threads : {64, 256}) {
    for (int wgs_per_cu : {1, 4}) {
      for (int lane_step : {1, 4}) {
        for (int sm : {0, 1, 3, 63}) {
          hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
          const int grid = 256 * wgs_per_cu;
          hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(wg_threads), 0, 0, 16, sm, lane_step, sink, cyc);
          hipEventRecord(a);
          hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(wg_threads), 0, 0, iters, sm, lane_step, sink, cyc);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
          const double waves_cu = wg_threads / 64.0 * wgs_per_cu;
          // per-CU time per wave instruction: kernel time / (iters * waves per CU)
          printf("%-22s waves/CU %4.0f  active lanes %2d  sharing %2d: %7.1f ns per wave-op per CU  (%6.1f cycles of wg 0 per op; %.1f us)\n", name,
                 waves_cu, 64 / lane_step, sm + 1, ms * 1e6 / (iters * waves_cu), (double)c / iters, ms * 1e3);
        }
      }
    }
  }
}

int main() {
  float* sink; long long* cyc;
  hipMalloc(&sink, 64); hipMalloc(&cyc, 64);
  run<0>("ds_add_f32", sink, cyc);
  run<1>("ds_add_u32", sink, cyc);
  run<2>("ds_add_u64", sink, cyc);
  run<3>("read+add+write b32", sink, cyc);
  run<4>("read+add+write b128", sink, cyc);
  run<5>("bpermute (shfl_down)", sink, cyc);
  run<6>("ds_add_rtn_f32", sink, cyc);
  run<7>("ds_pk_add_f16", sink, cyc);
  return 0;
}
