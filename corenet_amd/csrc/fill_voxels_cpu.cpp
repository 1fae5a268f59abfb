// fill_inside_voxels_cpu: the host twin of the flood fill, part of the product library.
// Reference: cc/module.cc:24-29, cc/fill_voxels_cpu.cc:158-183 (entry point: clone, one grid per
// parallel_for task) and :74-155 (raster-scan union-find over equal-occupancy voxels; every voxel
// whose component is not the virtual outside region -- which touches only the x==0 / y==0 / z==0
// faces -- is overwritten with 1, ALL OTHER VOXELS KEEP THEIR INPUT VALUE, so negative or
// non-binary values of reached empty voxels survive: SURVEY Q10; the GPU op writes strict {0,1}).
//
// Not a union-find here: the same bit-parallel closure as the HIP kernel, on the host.  A grid is
// packed to 1 bit / voxel ("empty" and "reached" bitmaps); a row is closed along x with carry
// arithmetic on 64-voxel words, and a worklist of dirty rows propagates reach to the four
// neighbouring rows (y +- 1, z +- 1) until nothing changes.  A monotone closure: the fixed point is
// the set of empty voxels 6-connected to the low faces, i.e. the reference's outside component.
// Grids are independent: one std::thread per chunk of grids.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/corenet_hip.h"

namespace {

typedef unsigned long long u64;

inline u64 spread_up(u64 e, u64 r) { return (((e + r) ^ e) & e) | r; }      // r flows to higher bits through runs of e

inline u64 reverse_bits(u64 v) {
  v = ((v >> 1) & 0x5555555555555555ull) | ((v & 0x5555555555555555ull) << 1);
  v = ((v >> 2) & 0x3333333333333333ull) | ((v & 0x3333333333333333ull) << 2);
  v = ((v >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((v & 0x0F0F0F0F0F0F0F0Full) << 4);
  return __builtin_bswap64(v);
}

// closes row r inside e along x (both directions, carries across words); true if r changed
inline bool close_row(const u64* e, u64* r, int wx) {
  bool changed = false;
  u64 carry = 0;
  for (int k = 0; k < wx; ++k) {
    u64 v = spread_up(e[k], r[k] | (carry & e[k] & 1ull));
    carry = v >> 63;
    if (v != r[k]) { r[k] = v; changed = true; }
  }
  carry = 0;
  for (int k = wx - 1; k >= 0; --k) {
    const u64 eb = reverse_bits(e[k]);
    u64 v = spread_up(eb, reverse_bits(r[k]) | (carry & eb & 1ull));
    carry = v >> 63;
    v = reverse_bits(v);
    if (v != r[k]) { r[k] = v; changed = true; }
  }
  return changed;
}

template <typename T>
void fill_one_grid(const T* in, T* out, int D, int H, int W) {
  const int wx = (W + 63) / 64;
  const int64_t rows = (int64_t)D * H;
  std::vector<u64> E((size_t)rows * wx, 0ull), R((size_t)rows * wx, 0ull);
  std::vector<int32_t> work;                 // dirty rows (z*H + y), LIFO
  std::vector<uint8_t> queued((size_t)rows, 0);
  work.reserve((size_t)rows);
  for (int64_t row = 0; row < rows; ++row) {
    const T* src = in + row * W;
    u64* e = &E[(size_t)row * wx];
    for (int x = 0; x < W; ++x)
      if (!(src[x] > (T)0)) e[x >> 6] |= 1ull << (x & 63);
    const int y = (int)(row % H), z = (int)(row / H);
    u64* r = &R[(size_t)row * wx];
    if (y == 0 || z == 0) std::memcpy(r, e, sizeof(u64) * wx);     // the y == 0 and z == 0 faces touch the outside
    else r[0] = e[0] & 1ull;                                       // and so does x == 0
    bool seeded = false;
    for (int k = 0; k < wx; ++k) seeded |= r[k] != 0ull;
    if (seeded) { work.push_back((int32_t)row); queued[(size_t)row] = 1; }
  }
  auto push = [&](int64_t row) {
    if (!queued[(size_t)row]) { queued[(size_t)row] = 1; work.push_back((int32_t)row); }
  };
  while (!work.empty()) {
    const int64_t row = work.back();
    work.pop_back();
    queued[(size_t)row] = 0;
    const int y = (int)(row % H), z = (int)(row / H);
    const u64* e = &E[(size_t)row * wx];
    u64* r = &R[(size_t)row * wx];
    // reach of the four neighbouring rows enters through this row's empty voxels, then spreads along x
    for (int k = 0; k < wx; ++k) {
      u64 nb = 0ull;
      if (y > 0) nb |= R[(size_t)(row - 1) * wx + k];
      if (y + 1 < H) nb |= R[(size_t)(row + 1) * wx + k];
      if (z > 0) nb |= R[(size_t)(row - H) * wx + k];
      if (z + 1 < D) nb |= R[(size_t)(row + H) * wx + k];
      r[k] |= nb & e[k];
    }
    close_row(e, r, wx);
    // a neighbouring row with an empty, not yet reached voxel next to a reached one of this row is dirty
    auto feeds = [&](int64_t other) {
      const u64* eo = &E[(size_t)other * wx];
      const u64* ro = &R[(size_t)other * wx];
      for (int k = 0; k < wx; ++k)
        if (r[k] & eo[k] & ~ro[k]) return true;
      return false;
    };
    if (y > 0 && feeds(row - 1)) push(row - 1);
    if (y + 1 < H && feeds(row + 1)) push(row + 1);
    if (z > 0 && feeds(row - H)) push(row - H);
    if (z + 1 < D && feeds(row + H)) push(row + H);
  }
  // fill_voxels_cpu.cc:150-154: only voxels outside the reached set are overwritten
  for (int64_t row = 0; row < rows; ++row) {
    const u64* r = &R[(size_t)row * wx];
    const T* src = in + row * W;
    T* dst = out + row * W;
    for (int x = 0; x < W; ++x) dst[x] = ((r[x >> 6] >> (x & 63)) & 1ull) ? src[x] : (T)1;
  }
}

template <typename T>
int fill_all(const T* in, T* out, int N, int D, int H, int W, int threads) {
  const int64_t S = (int64_t)D * H * W;
  if (threads < 1) threads = (int)std::max(1u, std::thread::hardware_concurrency());
  threads = std::min(threads, N);
  if (threads <= 1) {
    for (int n = 0; n < N; ++n) fill_one_grid(in + n * S, out + n * S, D, H, W);
    return CRN_OK;
  }
  std::vector<std::thread> pool;
  for (int tix = 0; tix < threads; ++tix)
    pool.emplace_back([=] {
      for (int n = tix; n < N; n += threads) fill_one_grid(in + n * S, out + n * S, D, H, W);
    });
  for (auto& th : pool) th.join();
  return CRN_OK;
}

}  // namespace

extern "C" int crn_fill_voxels_cpu(const void* grid, void* out, int dtype, int N, int D, int H, int W,
                                   int num_threads) {
  if (!grid || !out || N < 1 || D < 1 || H < 1 || W < 1) return CRN_EINVAL;
  if ((int64_t)D * H >= (int64_t)1 << 31) return CRN_EINVAL;
  switch (dtype) {
    case 0: return fill_all((const float*)grid, (float*)out, N, D, H, W, num_threads);
    case 1: return fill_all((const uint8_t*)grid, (uint8_t*)out, N, D, H, W, num_threads);
    case 2: return fill_all((const int32_t*)grid, (int32_t*)out, N, D, H, W, num_threads);
    case 3: return fill_all((const double*)grid, (double*)out, N, D, H, W, num_threads);
    case 4: return fill_all((const int64_t*)grid, (int64_t*)out, N, D, H, W, num_threads);
    case 5: return fill_all((const int16_t*)grid, (int16_t*)out, N, D, H, W, num_threads);
    case 6: return fill_all((const int8_t*)grid, (int8_t*)out, N, D, H, W, num_threads);
  }
  return CRN_EINVAL;
}
