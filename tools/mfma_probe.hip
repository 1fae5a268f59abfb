// Hardware probe of tools/ (NOT part of libcorenet_hip.so: built into tools/_build/libcrn_probe.so by
// corenet_amd.build.build_tools): a kernel of nothing but v_mfma_f32_16x16x32_bf16 in a fixed inline-assembly order, the
// "aggressor" of tools/mfma_neighbour.py and tests/test_kernels_gpu.py::test_ray_scatter_beside_mfma_neighbours.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define CRN_OK 0
#define CRN_EINVAL -1
#define CRN_CHECK_LAUNCH()                       \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)
typedef void* crnStream;

// ---- hardware probe (tools/mfma_neighbour.py) ------------------------------------------------------------
// A kernel that does nothing but issue v_mfma_f32_16x16x32_bf16 in a fixed pattern (inline assembly: the compiler's scheduler
// cannot reorder it), to be run beside a victim kernel on another stream (victim = the 64^3 ray scatter, 30 runs per mode):
//   mode 0   four independent accumulators, round robin                                                      exact
//   mode 1   ONE accumulator, four MFMAs per loop trip -- (a0,b0) (a1,b1) (a0,b1) (a1,b0) --, 8 idle cycles after each:
//            every MFMA waits for the one before it, a little late                                  wrong in 29 of 30
//   mode 33  mode 1 without the idle cycles (back to back)                                                    exact
//   mode 2   two accumulators interleaved; mode 4: the same with the A registers shared in pairs             exact
//   mode 3   one accumulator, two MFMAs per trip, 48 idle cycles after each; 12-15: 16-40; 20-28: 1-12       exact
//   mode 10  one accumulator, two MFMAs per trip back to back; 30 / 31: sharing the A / the B registers      exact
//   mode 32  the split-bf16 triple (a0,b0) (a0,b1) (a1,b0) on one accumulator, then on the next, back to back exact
//   modes 40-43  mode 1 with 1, 2, 4, 6 idle cycles after each MFMA                                          exact
//   modes 44-47  mode 1 with 12, 16, 24, 32 idle cycles                                             wrong in 28-29 of 30
//   mode 48  three MFMAs per trip, 8 idle cycles after each                                          wrong in 29 of 30
// i.e. three or more dependent MFMAs in a row with 8 or more idle cycles between them -- time in which the SIMD issues other waves'
// instructions while the accumulator is still owed -- do it, reliably; back to back, or two per loop trip, they do not.  The
// library's kernels have LDS reads and address arithmetic between the dependent MFMAs of an accumulator.
namespace {
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_probe_kernel(int mode, int iters, float* sink) {
  probe_bf16x8 a0, a1, b0, b1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0[i] = (__bf16)(float)((threadIdx.x + i) & 3); a1[i] = (__bf16)(float)((threadIdx.x * 3 + i) & 3);
    b0[i] = (__bf16)0.25f; b1[i] = (__bf16)0.125f;
  }
  probe_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#define CRN_PROBE_OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
  for (int it = 0; it < iters; ++it) {
    if (mode == 0)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %7, %1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %2, %5, %6, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %5, %7, %3\n\ts_nop 7" CRN_PROBE_OPS);
    else if (mode == 1)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\ts_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\ts_nop 7\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\ts_nop 7" CRN_PROBE_OPS);
    else if (mode == 2)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %7, %1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %6, %1\n\ts_nop 7" CRN_PROBE_OPS);
    else if (mode == 3)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" CRN_PROBE_OPS);
    else if (mode == 4)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %7, %1\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %7, %1\n\ts_nop 7" CRN_PROBE_OPS);
#define CRN_N8 "s_nop 7\n\t"
#define CRN_CHAIN(NOPS) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\t" NOPS "v_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\t" NOPS "s_nop 0" CRN_PROBE_OPS)
    else if (mode == 12) CRN_CHAIN(CRN_N8 CRN_N8);                       // modes 12 ... 15: one chain, 16 / 24 / 32 / 40 idle cycles
    else if (mode == 13) CRN_CHAIN(CRN_N8 CRN_N8 CRN_N8);
    else if (mode == 14) CRN_CHAIN(CRN_N8 CRN_N8 CRN_N8 CRN_N8);
    else if (mode == 15) CRN_CHAIN(CRN_N8 CRN_N8 CRN_N8 CRN_N8 CRN_N8);
    else if (mode == 20) CRN_CHAIN("s_nop 0\n\t");                       // modes 20 ... 27: one chain, 1 ... 8 idle cycles
    else if (mode == 21) CRN_CHAIN("s_nop 1\n\t");
    else if (mode == 22) CRN_CHAIN("s_nop 2\n\t");
    else if (mode == 23) CRN_CHAIN("s_nop 3\n\t");
    else if (mode == 24) CRN_CHAIN("s_nop 4\n\t");
    else if (mode == 25) CRN_CHAIN("s_nop 5\n\t");
    else if (mode == 26) CRN_CHAIN("s_nop 6\n\t");
    else if (mode == 27) CRN_CHAIN("s_nop 7\n\t");
    else if (mode == 28) CRN_CHAIN("s_nop 7\n\ts_nop 3\n\t");             // 12 idle cycles
    else if (mode == 30)      // one chain, consecutive MFMAs share the A registers
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\ts_nop 0" CRN_PROBE_OPS);
    else if (mode == 31)      // one chain, consecutive MFMAs share the B registers
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\ts_nop 0" CRN_PROBE_OPS);
    else if (mode == 32)      // the split-bf16 triple: (a0, b0), (a0, b1), (a1, b0) on one accumulator, then the same on the next
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %7, %1\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %6, %1\n\ts_nop 0" CRN_PROBE_OPS);
    else if (mode == 33)      // mode 1 without its idle cycles
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\t"
                   "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\ts_nop 0" CRN_PROBE_OPS);
#define CRN_CHAIN4(NOPS) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\t" NOPS "v_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\t" NOPS \
                                      "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\t" NOPS "v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\t" NOPS "s_nop 0" CRN_PROBE_OPS)
#define CRN_CHAIN3(NOPS) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %6, %0\n\t" NOPS "v_mfma_f32_16x16x32_bf16 %0, %5, %7, %0\n\t" NOPS \
                                      "v_mfma_f32_16x16x32_bf16 %0, %4, %7, %0\n\t" NOPS "s_nop 0" CRN_PROBE_OPS)
    else if (mode == 40) CRN_CHAIN4("s_nop 0\n\t");                      // modes 40 ... 47: mode 1 with 1, 2, 4, 6, 12, 16, 24, 32 idle cycles
    else if (mode == 41) CRN_CHAIN4("s_nop 1\n\t");
    else if (mode == 42) CRN_CHAIN4("s_nop 3\n\t");
    else if (mode == 43) CRN_CHAIN4("s_nop 5\n\t");
    else if (mode == 44) CRN_CHAIN4("s_nop 7\n\ts_nop 3\n\t");
    else if (mode == 45) CRN_CHAIN4("s_nop 7\n\ts_nop 7\n\t");
    else if (mode == 46) CRN_CHAIN4("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t");
    else if (mode == 47) CRN_CHAIN4("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t");
    else if (mode == 48) CRN_CHAIN3("s_nop 7\n\t");                      // three MFMAs per trip, 8 idle cycles after each
#undef CRN_CHAIN4
#undef CRN_CHAIN3
    else CRN_CHAIN("");                                                  // mode 10: one chain, back to back
#undef CRN_CHAIN
#undef CRN_N8
  }
#undef CRN_PROBE_OPS
  const probe_f32x4 r = c0 + c1 + c2 + c3;
  if (r[0] + r[1] + r[2] + r[3] == 123456.f) sink[0] = r[0];      // (keeps the accumulators live)
}
}  // namespace
extern "C" int crn_mfma_probe(int mode, int iters, int workgroups, float* sink, crnStream s) {
  if (!sink || iters < 1 || workgroups < 1 || mode < 0 || mode > 48) return CRN_EINVAL;
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)s, mode, iters, sink);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
