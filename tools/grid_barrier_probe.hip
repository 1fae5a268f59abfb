// Cost of a grid-wide barrier inside ONE launch on MI355X (all workgroups resident), as a persistent encoder-block kernel
// would use it between its phases: every thread stores 16 bytes written through (sc0 sc1), waits for its stores, the workgroup
// meets, thread 0 arrives with one relaxed agent-scope atomic and polls relaxed; after the barrier every thread reads a
// neighbour workgroup's value with a coherent (sc1) load and checks it.  Variants: flat (one counter), two-level (one counter per
// group of `gs` workgroups, the last arrival of a group arrives at the top counter), fence form (release / acquire at agent scope:
// what __threadfence() + atomics compile to).
//   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o tools/_build/grid_barrier_probe && tools/_build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct Sync { unsigned top; unsigned gen; unsigned abort; unsigned pad[13]; unsigned grp[64 * 16]; };

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>   // 0 flat, 1 two-level, 2 flat with fences
__device__ __forceinline__ bool grid_barrier(Sync* s, unsigned k, unsigned nwg, unsigned gs) {
  if (MODE == 2) __threadfence();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    bool release = false;
    if (MODE == 1) {
      const unsigned g = blockIdx.x / gs, ng = (nwg + gs - 1) / gs, mine = min(gs, nwg - g * gs);
      const unsigned t = __hip_atomic_fetch_add(&s->grp[g * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (k + 1) * mine - 1) {
        const unsigned t2 = __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        release = t2 == (k + 1) * ng - 1;
      }
    } else {
      const unsigned t = __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      release = t == (k + 1) * nwg - 1;
    }
    int good = 1;
    if (release) __hip_atomic_store(&s->gen, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      unsigned spins = 0;
      while (ld_relaxed(&s->gen) < k + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { __hip_atomic_store(&s->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break; }
      }
    }
    ok = good;
  }
  __syncthreads();
  if (MODE == 2) __threadfence();
  return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(Sync* s, float* buf, int phases, unsigned gs, int* errors, int payload4) {
  const unsigned nwg = gridDim.x;
  float4* mine = reinterpret_cast<float4*>(buf) + (size_t)blockIdx.x * 256 * payload4 + threadIdx.x;
  int bad = 0;
  for (int k = 0; k < phases; ++k) {
    const float v = (float)(k * 1000 + (int)blockIdx.x);
    for (int q = 0; q < payload4; ++q) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 w = {v, v + 1, v + 2, v + 3};
      if (MODE == 2) *reinterpret_cast<f32x4*>(mine + q * 256) = w;
      else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(mine + q * 256), "v"(w) : "memory");
    }
    if (!grid_barrier<MODE>(s, (unsigned)(2 * k), nwg, gs)) return;
    const unsigned other = (blockIdx.x + 37 * (k + 1)) % nwg;                  // a workgroup on another XCD
    const float4* theirs = reinterpret_cast<const float4*>(buf) + (size_t)other * 256 * payload4 + threadIdx.x;
    for (int q = 0; q < payload4; ++q) {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 r;
      if (MODE == 2) r = *reinterpret_cast<const f32x4*>(theirs + q * 256);
      else asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(theirs + q * 256) : "memory");
      bad += r[0] != (float)(k * 1000 + (int)other);
    }
    // (the next phase overwrites `mine` while others may still read it: a second barrier like the real use would have)
    if (!grid_barrier<MODE>(s, (unsigned)(2 * k + 1), nwg, gs)) return;
  }
  if (bad) atomicAdd(errors, bad);
}

template <int MODE>
void run(const char* name, int nwg, unsigned gs, int payload4) {
  Sync* s; float* buf; int* err;
  hipMalloc(&s, sizeof(Sync)); hipMalloc(&buf, (size_t)nwg * 256 * 16 * payload4); hipMalloc(&err, 4);
  const int phases = 200;
  float best = 1e9f;
  int herr = 0; Sync hs;
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(s, 0, sizeof(Sync)); hipMemset(err, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, s, buf, phases, gs, err, payload4);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hs, s, sizeof(Sync), hipMemcpyDeviceToHost);
  }
  printf("%-28s %4d workgroups, groups of %3u, %2d x 16 B per thread and phase: %6.2f us per barrier (2 per phase)  stale reads %d  abort %u\n",
         name, nwg, gs, payload4, best * 1e3 / (2 * phases), herr, hs.abort);
  hipFree(s); hipFree(buf); hipFree(err);
}

int main() {
  for (int payload4 : {1, 8}) {
    for (int nwg : {256, 512, 1024}) {
      run<0>("flat, write-through", nwg, 1, payload4);
      run<1>("two-level, write-through", nwg, 8, payload4);
      run<1>("two-level, write-through", nwg, 32, payload4);
      run<1>("two-level, write-through", nwg, 64, payload4);
      run<2>("flat, __threadfence", nwg, 1, payload4);
    }
  }
  return 0;
}
