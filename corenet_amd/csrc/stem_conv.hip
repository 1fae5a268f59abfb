// The encoder's stem on its own kernels: ZeroPad2d(3) + Conv2d(3 -> 64, 7 x 7, stride 2) (resnet50.py:122-124), forward
// (+ the partial sums of the BatchRenorm that follows, resnet50.py:125) and weight gradient.  The layer has no data
// gradient (its input is the image).
//
// Why not the generic engine (crn_conv_fwd on the 2 x 2 space-to-depth view, 4 x 4 window over 12 channels): that form
// multiplies 192 taps per position for 147 real ones and stages the image plane by plane through the engine's K-chunk
// loop -- 54 us forward and 64 us weight gradient for 1.2 GFLOP each, both on the step's critical chain (the weight
// gradient is the LAST thing a step waits for, the forward the first thing the next one needs).  Here a workgroup holds
// the whole 3 x 7 x 8 window (kw padded to 8: a k-step of the fp32 MFMA is 4 taps of one window row) in LDS next to the
// image patch of its 4 x 32 output tile; every A / B fragment address is lane constant + immediate.
//   GEMM view, forward:  D[position][n] += X[position][k] * W[k][n],  k = (c, kh, kw8): 42 k-steps of 4
//   weight gradient:     D[k][n] += X^T[k][position] * dY[position][n], positions of a tile in k-steps of 4
// Both use v_mfma_f32_16x16x4_f32 (the precision of the fp32 engine this layer ran on before).
// Weights / weight gradients are read / added in the PACKED layout of conv_geometry.stem_fwd ([12][16][64]:
// channel c*4 + rh*2 + rw, tap zh*4 + zw, kh + 1 = 2 zh + rh, kw + 1 = 2 zw + rw), so pack, un-pack, buckets and Adam
// around the layer stay what they are.
#include "crn_common.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr int kTH = 4, kTW = 32;              // output tile
constexpr int kPH = 2 * kTH + 5;              // patch rows (13)
constexpr int kPWf = 72;                      // patch row pitch, forward (69 columns used; floats)
constexpr int kPWg = 80;                      // ... weight gradient (two rows of 8 taps land on disjoint banks)
constexpr int kKR = 3 * 7 * 8;                // window rows of the K dimension (kw padded to 8)
constexpr int kWP = 80;                       // LDS pitch of a weight row (64 columns): lane groups kk, kk + 1 on disjoint banks
constexpr int kDP = 84;                       // LDS pitch of a dy position row (64 columns)
constexpr int kThreads = 256;

struct StemGeom {
  const float* img; const float* w; const float* bias; float* y; double* ws;
  const float* dy; float* dw;
  int B, H, W, H1, W1, tilesH, tilesW, ntiles;
};

// packed row ([12][16]) of window tap (c, kh, kw), kw < 7
__device__ __forceinline__ int packed_row(int c, int kh, int kw) {
  const int rh = (kh + 1) & 1, zh = (kh + 1) >> 1, rw = (kw + 1) & 1, zw = (kw + 1) >> 1;
  return (c * 4 + rh * 2 + rw) * 16 + zh * 4 + zw;
}

__device__ __forceinline__ void tile_of(const StemGeom& g, int tile, int& b, int& oh0, int& ow0) {
  const int tw = tile % g.tilesW; tile /= g.tilesW;
  const int th = tile % g.tilesH; b = tile / g.tilesH;
  oh0 = th * kTH; ow0 = tw * kTW;
}

// the image patch of a tile: rows 2 oh0 - 3 ..., columns 2 ow0 - 3 ... (zeros outside the image: ZeroPad2d(3))
template <int PW, int NE>
__device__ __forceinline__ void patch_load(const StemGeom& g, int b, int oh0, int ow0, int tid, float (&pv)[NE]) {
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = tid + j * kThreads;
    const int c = e / (kPH * PW), rem = e - c * (kPH * PW), row = rem / PW, col = rem - row * PW;
    const int gh = 2 * oh0 + row - 3, gw = 2 * ow0 + col - 3;
    const bool ok = e < 3 * kPH * PW && (unsigned)gh < (unsigned)g.H && (unsigned)gw < (unsigned)g.W;
    pv[j] = ok ? g.img[((int64_t)(b * 3 + c) * g.H + gh) * g.W + gw] : 0.f;
  }
}
template <int PW, int NE>
__device__ __forceinline__ void patch_store(float* pl, int tid, const float (&pv)[NE]) {
#pragma unroll
  for (int j = 0; j < NE; ++j) {
    const int e = tid + j * kThreads;
    if (e < 3 * kPH * PW) pl[e] = pv[j];
  }
}

// ---------------------------------------------------------------- forward ---------------------------------------
// workgroup = one 4 x 32 tile of one sample, 4 waves; wave w owns output row w: 2 sub-tiles of 16 columns x 4 blocks of
// 16 channels.  ws != nullptr: sum(y), sum(y^2) of the tile per channel -> ws[(n * gridDim.x + blockIdx.x) * 2] (the
// partial-sum layout of bn_finalize_kernel, csrc/batch_renorm.hip): the statistics pass over y is not needed.
__global__ __launch_bounds__(kThreads) void stem_fwd_kernel(StemGeom g) {
  __shared__ __attribute__((aligned(16))) float wl[kKR * kWP];
  __shared__ float pl[3 * kPH * kPWf];
  crn_kernarg_touch(g);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kk = lane >> 4;
  int b, oh0, ow0;
  tile_of(g, blockIdx.x, b, oh0, ow0);
  // every global load is issued before the first LDS write
  constexpr int NWV = (kKR * 16 + kThreads - 1) / kThreads;      // float4 of the window per thread (10.5)
  constexpr int NPE = (3 * kPH * kPWf + kThreads - 1) / kThreads;
  f32x4 wv[NWV];
  float pv[NPE];
#pragma unroll
  for (int j = 0; j < NWV; ++j) {
    const int f = tid + j * kThreads;
    const int k = f >> 4, n4 = f & 15;
    const int ck = k >> 3, kw = k & 7, c = ck / 7, kh = ck - c * 7;
    const bool ok = f < kKR * 16 && kw < 7;
    wv[j] = ok ? *reinterpret_cast<const f32x4*>(g.w + (int64_t)packed_row(c, kh, kw) * 64 + n4 * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  patch_load<kPWf>(g, b, oh0, ow0, tid, pv);
  float bsv[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) bsv[nb] = g.bias ? g.bias[nb * 16 + i16] : 0.f;
#pragma unroll
  for (int j = 0; j < NWV; ++j) {
    const int f = tid + j * kThreads;
    if (f < kKR * 16) *reinterpret_cast<f32x4*>(wl + (f >> 4) * kWP + (f & 15) * 4) = wv[j];
  }
  patch_store<kPWf>(pl, tid, pv);
  __syncthreads();
  f32x4 acc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[h][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* pa = pl + (2 * wave) * kPWf + 2 * i16 + kk;
  const float* pb = wl + kk * kWP + i16;
#pragma unroll
  for (int s = 0; s < kKR / 4; ++s) {
    const int c = s / 14, r2 = s - c * 14, kh = r2 >> 1, h4 = r2 & 1;
    const int ao = (c * kPH + kh) * kPWf + 4 * h4;
    const float a0 = pa[ao], a1 = pa[ao + 32];
    float bv[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) bv[nb] = pb[4 * s * kWP + nb * 16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv[nb], acc[0][nb], 0, 0, 0);
      acc[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv[nb], acc[1][nb], 0, 0, 0);
    }
  }
  // D rows kk*4 .. kk*4+3 = 4 consecutive columns of the sub-tile, D column i16 = channel
  const int oh = oh0 + wave;
  float s1[4], s2[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    s1[nb] = s2[nb] = 0.f;
    const int n = nb * 16 + i16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ow = ow0 + 16 * h + 4 * kk;
      if (oh < g.H1 && ow < g.W1) {                  // (W1 % 4 == 0: a float4 is inside the row or outside it)
        const f32x4 v = acc[h][nb] + bsv[nb];
        *reinterpret_cast<f32x4*>(g.y + ((int64_t)(b * 64 + n) * g.H1 + oh) * g.W1 + ow) = v;
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[nb] += v[r]; s2[nb] += v[r] * v[r]; }
      }
    }
  }
  if (!g.ws) return;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    s1[nb] += __shfl_xor(s1[nb], 16); s1[nb] += __shfl_xor(s1[nb], 32);
    s2[nb] += __shfl_xor(s2[nb], 16); s2[nb] += __shfl_xor(s2[nb], 32);
  }
  __syncthreads();                                   // every wave is done with the window image: it becomes the scratch
  float* red = wl;                                   // [4 waves][64 channels][2]
  if (kk == 0) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      red[(wave * 64 + nb * 16 + i16) * 2 + 0] = s1[nb];
      red[(wave * 64 + nb * 16 + i16) * 2 + 1] = s2[nb];
    }
  }
  __syncthreads();
  if (tid < 64) {
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += (double)red[(w * 64 + tid) * 2]; q += (double)red[(w * 64 + tid) * 2 + 1]; }
    double* o = g.ws + ((int64_t)tid * gridDim.x + blockIdx.x) * 2;
    o[0] = a; o[1] = q;
  }
}

// ---------------------------------------------------------------- weight gradient ------------------------------
// workgroup = 4 waves, wave w owns output channels 16 w .. 16 w + 15 and all 11 blocks of 16 window rows (168 of 176 real);
// it walks its tiles (blockIdx.x, + gridDim.x, ...) with the accumulators in registers and adds them to the packed
// gradient once at the end (fire-and-forget atomics: dw is zero or holds earlier contributions).
__global__ __launch_bounds__(kThreads) void stem_wgrad_kernel(StemGeom g) {
  __shared__ float pl[3 * kPH * kPWg];
  __shared__ float dl[kTH * kTW * kDP];
  crn_kernarg_touch(g);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kk = lane >> 4;
  constexpr int NMB = (kKR + 15) / 16;               // 11
  int abase[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
    const int T = min(mb * 16 + i16, kKR - 1);
    const int ck = T >> 3, kw = T & 7, c = ck / 7, kh = ck - c * 7;
    abase[mb] = (c * kPH + kh) * kPWg + kw + 2 * kk;
  }
  const int bbase = kk * kDP + wave * 16 + i16;
  f32x4 acc[NMB];
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) acc[mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NPE = (3 * kPH * kPWg + kThreads - 1) / kThreads;
  constexpr int NDV = kTH * kTW * 64 / 4 / kThreads;  // 8 float4 of dy per thread
  for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
    int b, oh0, ow0;
    tile_of(g, tile, b, oh0, ow0);
    float pv[NPE];
    f32x4 dv[NDV];
    patch_load<kPWg>(g, b, oh0, ow0, tid, pv);
#pragma unroll
    for (int j = 0; j < NDV; ++j) {
      const int f = tid + j * kThreads;
      const int n = (f >> 9) * 16 + (f & 15), q4 = (f >> 4) & 31;
      const int oh = oh0 + (q4 >> 3), ow = ow0 + (q4 & 7) * 4;
      dv[j] = (oh < g.H1 && ow < g.W1) ? *reinterpret_cast<const f32x4*>(g.dy + ((int64_t)(b * 64 + n) * g.H1 + oh) * g.W1 + ow)
                                       : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                 // the previous tile's fragments are read
    patch_store<kPWg>(pl, tid, pv);
#pragma unroll
    for (int j = 0; j < NDV; ++j) {
      const int f = tid + j * kThreads;
      const int n = (f >> 9) * 16 + (f & 15), q4 = (f >> 4) & 31;
      float* d = dl + (q4 * 4) * kDP + n;            // position (row q4 >> 3, column (q4 & 7) * 4 + i) = q4 * 4 + i
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i * kDP] = dv[j][i];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kTH * kTW / 4; ++q) {        // k-step q: positions 4 q + kk = (row q >> 3, column 4 (q & 7) + kk)
      const int pr = q >> 3, qc = q & 7;
      const float bv = dl[bbase + 4 * q * kDP];
      const int ao = 2 * pr * kPWg + 8 * qc;
#pragma unroll
      for (int mb = 0; mb < NMB; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pl[abase[mb] + ao], bv, acc[mb], 0, 0, 0);
    }
  }
  // D row kk*4 + r = window row mb*16 + kk*4 + r, D column i16 = channel 16 wave + i16
  const int n = wave * 16 + i16;
#pragma unroll
  for (int mb = 0; mb < NMB; ++mb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int T = mb * 16 + kk * 4 + r;
      const int ck = T >> 3, kw = T & 7, c = ck / 7, kh = ck - c * 7;
      if (T < kKR && kw < 7) atomicAdd(g.dw + (int64_t)packed_row(c, kh, kw) * 64 + n, acc[mb][r]);
    }
  }
}

bool stem_shape_ok(int B, int H, int W) {
  // (even extents: H1 = H / 2 exactly; W1 % 4 == 0: float4 rows of y / dy; 32-bit tile count)
  return B >= 1 && H >= 8 && W >= 8 && (H & 1) == 0 && (W & 7) == 0 && (int64_t)B * H * W < ((int64_t)1 << 30);
}
StemGeom stem_geom(int B, int H, int W) {
  StemGeom g{};
  g.B = B; g.H = H; g.W = W; g.H1 = H / 2; g.W1 = W / 2;
  g.tilesH = crn_cdiv(g.H1, kTH); g.tilesW = crn_cdiv(g.W1, kTW);
  g.ntiles = B * g.tilesH * g.tilesW;
  return g;
}
}  // namespace

extern "C" size_t crn_stem_conv_parts(int B, int H, int W) {
  if (!stem_shape_ok(B, H, W)) return 0;
  return (size_t)stem_geom(B, H, W).ntiles;
}

extern "C" int crn_stem_conv_fwd(const float* img, int B, int H, int W, const float* w_packed, const float* bias,
                                 float* y, double* stats_ws, size_t ws_bytes, crnStream stream) {
  CRN_ENTRY(stream);
  if (!img || !w_packed || !y || !stem_shape_ok(B, H, W)) return CRN_EINVAL;
  if ((((uintptr_t)w_packed) | ((uintptr_t)y)) & 15) return CRN_EINVAL;
  StemGeom g = stem_geom(B, H, W);
  g.img = img; g.w = w_packed; g.bias = bias; g.y = y; g.ws = stats_ws;
  if (stats_ws && ws_bytes < (size_t)64 * g.ntiles * 2 * sizeof(double)) return CRN_ENOMEM;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)g.ntiles), dim3(kThreads), 0, (hipStream_t)stream, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_stem_conv_wgrad(const float* img, int B, int H, int W, const float* dy, float* dw_packed,
                                   crnStream stream) {
  CRN_ENTRY(stream);
  if (!img || !dy || !dw_packed || !stem_shape_ok(B, H, W)) return CRN_EINVAL;
  if (((uintptr_t)dy) & 15) return CRN_EINVAL;
  if (crn_deterministic()) return CRN_EINVAL;        // (order-independent sums: the caller keeps crn_conv_wgrad)
  StemGeom g = stem_geom(B, H, W);
  g.img = img; g.dy = dy; g.dw = dw_packed;
  static const int blocks_env = getenv("CRN_STEM_WG_BLOCKS") ? atoi(getenv("CRN_STEM_WG_BLOCKS")) : 256;
  const int blocks = std::max(1, std::min(g.ntiles, blocks_env));
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
