// fill_inside_voxels: flood fill of the empty region connected to the
// x==0 / y==0 / z==0 faces; everything else becomes 1.
// Reference: cc/fill_voxels_gpu.cu:96-171 (lock-free union-find, 2 kernels,
// 16 B/voxel parent array), cc/fill_voxels_cpu.cc:74-183.  Semantics kept
// bit-exact: 6-connectivity, the virtual background node touches
// only the LOW faces, output strictly {0,1}.
//
// MI355X design: not a union-find.  The grid is bit-packed (1 bit/voxel: a
// 128^3 grid is 256 KiB), so a z-slab of the "empty" and "reached" bitmaps lives
// entirely in the 160 KiB LDS of one CU.  Propagation along x is word-parallel
// carry arithmetic (64 voxels per add), along y/z it is a bitwise AND/OR of
// neighbouring rows.  Slabs exchange halo planes through HBM between rounds;
// the propagation is a monotone closure, so any interleaving converges to the
// same fixed point = the reference's connected-component answer.
// HBM traffic: read grid once, write result once (+ 2 bits/voxel of bitmaps).
#include "crn_common.h"
#include <algorithm>

namespace {

typedef unsigned long long u64;

template <typename T>
__global__ __launch_bounds__(256) void fill_pack_kernel(const T* grid, u64* E, u64* R, int D, int H, int W,
                                                        int WX, int64_t nwords) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t wi = wave0; wi < nwords; wi += nwaves) {
    const int k = (int)(wi % WX);
    const int64_t row = wi / WX;              // (n*D + z)*H + y
    const int y = (int)(row % H);
    const int z = (int)((row / H) % D);
    const int x = k * 64 + lane;
    bool empty = false;
    if (x < W) empty = !(grid[row * W + x] > (T)0);
    const u64 e = __ballot(empty);
    if (lane == 0) {
      E[wi] = e;
      R[wi] = (y == 0 || z == 0) ? e : (k == 0 ? (e & 1ull) : 0ull);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fill_unpack_kernel(const u64* E, const u64* R, T* out, int W, int WX,
                                                          int64_t nwords) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t wi = wave0; wi < nwords; wi += nwaves) {
    const int k = (int)(wi % WX);
    const int64_t row = wi / WX;
    const int x = k * 64 + lane;
    const u64 outside = E[wi] & R[wi];
    if (x < W) out[row * W + x] = ((outside >> lane) & 1ull) ? (T)0 : (T)1;
  }
}

// bits of r (subset of e) spread towards higher bit positions through runs of 1s in e
__device__ __forceinline__ u64 fill_up(u64 e, u64 r) { return (((e + r) ^ e) & e) | r; }

template <int WX>
__device__ __forceinline__ bool relax_row(const u64* e, u64* r) {
  // r |= x-closure of r inside e, both directions, with carries across the row's words
  bool changed = false;
  u64 carry = 0;
#pragma unroll
  for (int k = 0; k < WX; ++k) {
    u64 v = r[k] | (carry & e[k] & 1ull);
    v = fill_up(e[k], v);
    carry = v >> 63;
    if (v != r[k]) { r[k] = v; changed = true; }
  }
  carry = 0;
#pragma unroll
  for (int k = WX - 1; k >= 0; --k) {
    const u64 eb = __brevll(e[k]);
    u64 v = __brevll(r[k]) | (carry & eb & 1ull);
    v = fill_up(eb, v);
    carry = v >> 63;
    v = __brevll(v);
    if (v != r[k]) { r[k] = v; changed = true; }
  }
  return changed;
}

constexpr int STRIP = 8;

template <int WX>
__global__ __launch_bounds__(512) void fill_sweep_kernel(const u64* E, u64* R, int D, int H, int zs,
                                                         int nslabs, int* flags, int round) {
  extern __shared__ __attribute__((aligned(16))) u64 sm[];
  if (round > 0 && flags[round - 1] == 0) return;      // converged in an earlier round
  const int slab = blockIdx.x % nslabs, n = blockIdx.x / nslabs;
  const int z0 = slab * zs, nz = min(zs, D - z0);
  const int rowsz = H * WX;
  u64* El = sm;                          // [nz][H][WX]
  u64* Rl = sm + (size_t)zs * rowsz;     // [nz+2][H][WX], plane 0 / nz+1 = halos
  const u64* Eg = E + ((int64_t)n * D + z0) * rowsz;
  u64* Rg = R + ((int64_t)n * D + z0) * rowsz;
  for (int i = threadIdx.x; i < nz * rowsz; i += blockDim.x) { El[i] = Eg[i]; Rl[rowsz + i] = Rg[i]; }
  for (int i = threadIdx.x; i < rowsz; i += blockDim.x) {
    Rl[i] = z0 > 0 ? Rg[i - rowsz] : 0ull;
    Rl[(nz + 1) * rowsz + i] = (z0 + nz < D) ? Rg[nz * rowsz + i] : 0ull;
  }
  __syncthreads();
  const int nstrips = (H + STRIP - 1) / STRIP;
  const int units = nz * nstrips;
  int any = 0;
  for (int iter = 0; iter < 100000; ++iter) {
    int changed = 0;
    for (int u = threadIdx.x; u < units; u += blockDim.x) {
      const int p = u / nstrips, y0 = (u % nstrips) * STRIP, y1 = min(H, y0 + STRIP);
      const u64* el = El + (size_t)p * rowsz;
      u64* rc = Rl + (size_t)(p + 1) * rowsz;
      const u64* rlo = Rl + (size_t)p * rowsz;
      const u64* rhi = Rl + (size_t)(p + 2) * rowsz;
      u64 prev[WX], e[WX], r[WX];
      // +y sweep
#pragma unroll
      for (int k = 0; k < WX; ++k) prev[k] = y0 > 0 ? rc[(y0 - 1) * WX + k] : 0ull;
      for (int y = y0; y < y1; ++y) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < WX; ++k) {
          e[k] = el[y * WX + k];
          const u64 old = rc[y * WX + k];
          const u64 nb = prev[k] | rlo[y * WX + k] | rhi[y * WX + k] | (y + 1 < H ? rc[(y + 1) * WX + k] : 0ull);
          r[k] = old | (nb & e[k]);
          ch |= (r[k] != old);
        }
        ch |= relax_row<WX>(e, r);
        if (ch) {
#pragma unroll
          for (int k = 0; k < WX; ++k) rc[y * WX + k] = r[k];
          changed = 1;
        }
#pragma unroll
        for (int k = 0; k < WX; ++k) prev[k] = r[k];
      }
      // -y sweep
#pragma unroll
      for (int k = 0; k < WX; ++k) prev[k] = y1 < H ? rc[y1 * WX + k] : 0ull;
      for (int y = y1 - 1; y >= y0; --y) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < WX; ++k) {
          e[k] = el[y * WX + k];
          const u64 old = rc[y * WX + k];
          const u64 nb = prev[k] | rlo[y * WX + k] | rhi[y * WX + k] | (y > 0 ? rc[(y - 1) * WX + k] : 0ull);
          r[k] = old | (nb & e[k]);
          ch |= (r[k] != old);
        }
        ch |= relax_row<WX>(e, r);
        if (ch) {
#pragma unroll
          for (int k = 0; k < WX; ++k) rc[y * WX + k] = r[k];
          changed = 1;
        }
#pragma unroll
        for (int k = 0; k < WX; ++k) prev[k] = r[k];
      }
    }
    if (!__syncthreads_or(changed)) break;
    any = 1;
  }
  if (any) {
    for (int i = threadIdx.x; i < nz * rowsz; i += blockDim.x) Rg[i] = Rl[rowsz + i];
    if (threadIdx.x == 0) atomicOr(&flags[round], 1);
  }
}

constexpr int kMaxRounds = 4096;
constexpr size_t kSweepLds = 144 * 1024;

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename T>
int run_fill(const T* grid, T* out, int N, int D, int H, int W, void* ws, hipStream_t st) {
  const int WX = (W + 63) / 64;
  if (WX > 8) return CRN_EINVAL;
  const int64_t nwords = (int64_t)N * D * H * WX;
  u64* E = reinterpret_cast<u64*>(ws);
  u64* R = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws) + align256((size_t)nwords * 8));
  int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + 2 * align256((size_t)nwords * 8));
  CRN_HIP(hipMemsetAsync(flags, 0, kMaxRounds * sizeof(int), st));
  const unsigned pk_blocks = (unsigned)std::min<int64_t>(std::max<int64_t>(1, (nwords + 3) / 4), 8192);
  hipLaunchKernelGGL(fill_pack_kernel<T>, dim3(pk_blocks), dim3(256), 0, st, grid, E, R, D, H, W, WX, nwords);
  CRN_CHECK_LAUNCH();
  const size_t plane = (size_t)H * WX * 8;
  int zs = (int)std::min<size_t>((size_t)D, (kSweepLds / plane - 2) / 2);
  if (zs < 1) return CRN_EINVAL;
  const int nslabs = (D + zs - 1) / zs;
  zs = (D + nslabs - 1) / nslabs;                       // balance the slabs
  const size_t lds = (size_t)(2 * zs + 2) * plane;
  auto launch = [&](int round) -> int {
    dim3 grid_(nslabs * N);
#define CRN_SWEEP(K)                                                                                           \
  case K: {                                                                                                    \
    if (lds > 65536)                                                                                           \
      CRN_HIP(hipFuncSetAttribute((const void*)fill_sweep_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)lds));                                                                  \
    hipLaunchKernelGGL(fill_sweep_kernel<K>, grid_, dim3(512), lds, st, E, R, D, H, zs, nslabs, flags, round);  \
  } break;
    switch (WX) { CRN_SWEEP(1) CRN_SWEEP(2) CRN_SWEEP(3) CRN_SWEEP(4) CRN_SWEEP(5) CRN_SWEEP(6) CRN_SWEEP(7) CRN_SWEEP(8) }
#undef CRN_SWEEP
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  };
  int round = 0;
  const int batch = nslabs == 1 ? 1 : 4;
  for (;;) {
    for (int i = 0; i < batch && round < kMaxRounds; ++i, ++round) {
      const int rc = launch(round);
      if (rc != CRN_OK) return rc;
    }
    if (nslabs == 1) break;                 // a single slab converges inside one launch
    int last = 1;
    CRN_HIP(hipMemcpyAsync(&last, flags + round - 1, sizeof(int), hipMemcpyDeviceToHost, st));
    CRN_HIP(hipStreamSynchronize(st));
    if (last == 0) break;
    if (round >= kMaxRounds) return CRN_ENOCONV;
  }
  hipLaunchKernelGGL(fill_unpack_kernel<T>, dim3(pk_blocks), dim3(256), 0, st, E, R, out, W, WX, nwords);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

}  // namespace

extern "C" size_t crn_fill_voxels_workspace_bytes(int N, int D, int H, int W) {
  const int WX = (W + 63) / 64;
  const size_t nwords = (size_t)N * D * H * WX;
  return 2 * align256(nwords * 8) + kMaxRounds * sizeof(int) + 256;
}

extern "C" int crn_fill_voxels(const void* grid, void* out, int dtype, int N, int D, int H, int W,
                               void* workspace, size_t workspace_bytes, crnStream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!grid || !out || N < 1 || D < 1 || H < 1 || W < 1) return CRN_EINVAL;
  if (workspace_bytes < crn_fill_voxels_workspace_bytes(N, D, H, W)) return CRN_ENOMEM;
  switch (dtype) {
    case 0: return run_fill<float>((const float*)grid, (float*)out, N, D, H, W, workspace, st);
    case 1: return run_fill<uint8_t>((const uint8_t*)grid, (uint8_t*)out, N, D, H, W, workspace, st);
    case 2: return run_fill<int32_t>((const int32_t*)grid, (int32_t*)out, N, D, H, W, workspace, st);
    case 3: return run_fill<double>((const double*)grid, (double*)out, N, D, H, W, workspace, st);
    case 4: return run_fill<int64_t>((const int64_t*)grid, (int64_t*)out, N, D, H, W, workspace, st);
    case 5: return run_fill<int16_t>((const int16_t*)grid, (int16_t*)out, N, D, H, W, workspace, st);
    case 6: return run_fill<int8_t>((const int8_t*)grid, (int8_t*)out, N, D, H, W, workspace, st);
  }
  return CRN_EINVAL;
}
