// LDS-DMA semantics probe (gfx950): where do the 16 bytes of lane l land for M0 = base, with and without an instruction
// offset?   hipcc --offload-arch=gfx950 -O2 tools/dma_probe.hip -o tools/_build/dma_probe && tools/_build/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned* src, unsigned* dst, int mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned long long a = (unsigned long long)src;
  rsrc_t rs = {(int)(unsigned)a, (int)((a >> 32) & 0xFFFFu), (int)0x7FFFFFFFu, 0x00020000};
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)smem;
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 256u);
  const unsigned off = threadIdx.x * 32u;
  if (threadIdx.x < 64) {
    if (mode == 0)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(m0v), "v"(off), "s"(rs) : "memory");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:16 lds" :: "s"(m0v), "v"(off), "s"(rs) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) dst[i] = smem[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;      // dword i holds i
  unsigned *s, *d;
  hipMalloc(&s, 4096 * 4); hipMalloc(&d, 1024 * 4);
  hipMemcpy(s, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    k<<<1, 128, 4096>>>(s, d, mode);
    std::vector<unsigned> o(1024);
    hipMemcpy(o.data(), d, 1024 * 4, hipMemcpyDeviceToHost);
    printf("mode %d (inst offset %d): LDS dwords that changed (index: value):\n", mode, mode * 16);
    int shown = 0, first = -1, last = -1;
    for (int i = 0; i < 1024; ++i) if (o[i] != 0xdeadbeefu) { if (first < 0) first = i; last = i; if (shown < 12) { printf("  [%d]=%u", i, o[i]); ++shown; } }
    printf("\n  first %d last %d ; lane1's dwords at [%d..]: %u %u %u %u\n", first, last, first + 4, o[first + 4], o[first + 5], o[first + 6], o[first + 7]);
  }
  return 0;
}
