// Small HBM-bound kernels around the conv engine: image preprocessing, the
// stem's BN+ReLU+maxpool, global average, the latent Linear layer, weight
// (un)packing gathers, Adam.  References cited per kernel.
#include "crn_common.h"
#include <cstdlib>
#include <algorithm>
#include <cmath>

namespace {

// resnet50.py:189-204: u8 RGB -> f32, flip to BGR, ADD the ImageNet means
__global__ void preprocess_kernel(const uint8_t* img, int64_t HW, float* out, int64_t total) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t s = e % HW;
  const int c = (int)((e / HW) % 3);
  const int64_t b = e / (3 * HW);
  const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
  out[e] = (float)img[(b * 3 + (2 - c)) * HW + s] + mean;
}

// resnet50.py:126-131: BatchRenorm -> ReLU -> ZeroPad2d(1) -> MaxPool2d(3, stride 2)
__global__ void bn_relu_maxpool_fwd_kernel(const float* x, const float* scale, const float* shift,
                                           int C, int H, int W, int Ho, int Wo, float* y,
                                           int32_t* argmax, int64_t total) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int ow = (int)(e % Wo);
  const int oh = (int)((e / Wo) % Ho);
  const int64_t bc = e / ((int64_t)Wo * Ho);
  const int c = (int)(bc % C);
  const float sc = scale[c], sh = shift[c];
  const float* p = x + bc * (int64_t)H * W;
  float best = -INFINITY;
  int bi = -1;
  // all nine loads are issued up front from clamped addresses (inside the bounds check they were nine memory
  // latencies one after the other); padding positions then take the value 0 like before
  float raw[9];
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int ih = min(max(2 * oh - 1 + t9 / 3, 0), H - 1), iw = min(max(2 * ow - 1 + t9 % 3, 0), W - 1);
    raw[t9] = p[ih * W + iw];
  }
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int ih = 2 * oh - 1 + t9 / 3, iw = 2 * ow - 1 + t9 % 3;
    const bool in = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const float v = in ? fmaxf(raw[t9] * sc + sh, 0.f) : 0.f;           // the zero pad
    const int idx = in ? ih * W + iw : -1;
    if (v > best) { best = v; bi = idx; }
  }
  y[e] = best;
  argmax[e] = (best > 0.f) ? bi : -1;   // zero-valued maxima carry no gradient through the ReLU
}

// gather form of the max-pool backward: deterministic, no atomics, no zero-fill
__global__ void bn_relu_maxpool_bwd_kernel(const float* dy, const int32_t* argmax, int H, int W,
                                           int Ho, int Wo, float* dx, int64_t total) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int iw = (int)(e % W);
  const int ih = (int)((e / W) % H);
  const int64_t bc = e / ((int64_t)W * H);
  const int idx = ih * W + iw;
  const float* g = dy + bc * (int64_t)Ho * Wo;
  const int32_t* am = argmax + bc * (int64_t)Ho * Wo;
  float s = 0.f;
  // windows covering row ih: 2*oh-1 <= ih <= 2*oh+1 (at most 2 x 2; loaded from clamped addresses up front)
  int amv[4];
  float gv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oh = min(ih / 2 + (q >> 1), Ho - 1), ow = min(iw / 2 + (q & 1), Wo - 1);
    amv[q] = am[oh * Wo + ow];
    gv[q] = g[oh * Wo + ow];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oh = ih / 2 + (q >> 1), ow = iw / 2 + (q & 1);
    const bool in = oh <= (ih + 1) / 2 && ow <= (iw + 1) / 2 && oh < Ho && ow < Wo;
    if (in && amv[q] == idx) s += gv[q];
  }
  dx[e] = s;
}

// resnet50.py:183: avg = relu(x_pre).mean over H*W ; one wave per (b,c)
__global__ void relu_mean_fwd_kernel(const float* x, int C, int64_t S, int64_t sB, float* avg, int BC) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= BC) return;
  const int b = wid / C, c = wid % C;
  const float* p = x + (int64_t)b * sB + (int64_t)c * S;
  double s = 0.0;
  for (int64_t i = lane; i < S; i += 64) s += (double)fmaxf(p[i], 0.f);
  s = crn_wave_sum(s);
  if (lane == 0) avg[wid] = (float)(s / (double)S);
}

__global__ void relu_mean_bwd_kernel(const float* x, const float* davg, int C, int64_t S, int64_t sB,
                                     float* dx, int64_t sBdx, int accumulate, int64_t total) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t s = e % S;
  const int c = (int)((e / S) % C);
  const int64_t b = e / (S * C);
  const float g = x[b * sB + c * S + s] > 0.f ? davg[b * C + c] / (float)S : 0.f;
  float* o = dx + b * sBdx + c * S + s;
  *o = accumulate ? *o + g : g;
}

// nn.Linear (reconstruction_decoder.py:49): one wave per (b,n)
__global__ void linear_fwd_kernel(const float* x, const float* w, const float* bias, int B, int K,
                                  int N, float* y, int ldy) {
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wid >= B * N) return;
  const int b = wid / N, n = wid % N;
  float s = 0.f;
#pragma unroll 8                                         // (the loads of 8 steps in flight: the loop is a latency chain)
  for (int k = lane; k < K; k += 64) s += x[(int64_t)b * K + k] * w[(int64_t)n * K + k];
  s = crn_wave_sum(s);
  if (lane == 0) y[(int64_t)b * ldy + n] = s + (bias ? bias[n] : 0.f);
}
__global__ void linear_bwd_dx_kernel(const float* w, const float* dy, int lddy, int B, int K, int N, float* dx) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * K) return;
  const int b = e / K, k = e % K;
  float s = 0.f;
#pragma unroll 16
  for (int n = 0; n < N; ++n) s += dy[(int64_t)b * lddy + n] * w[(int64_t)n * K + k];
  dx[e] = s;
}
__global__ void linear_bwd_dw_kernel(const float* x, const float* dy, int lddy, int B, int K, int N,
                                     float* dw, float* db) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * K) return;
  const int n = e / K, k = e % K;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dy[(int64_t)b * lddy + n] * x[(int64_t)b * K + k];
  dw[e] = s;
  if (k == 0 && db) {
    float t = 0.f;
    for (int b = 0; b < B; ++b) t += dy[(int64_t)b * lddy + n];
    db[n] = t;
  }
}

__global__ void fill_offset_kernel(float* x, int64_t sB, int64_t S, int c0, const float* offset, int64_t total) {
  const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int64_t s = e % S;
  const int j = (int)((e / S) % 3);
  const int64_t b = e / (3 * S);
  x[b * sB + (int64_t)(c0 + j) * S + s] = offset[b * 3 + j];
}

// layer matrices of the ray-traced skips, layer_mats[s][b] = v2s[b] . scale(f_s, f_s, f_s) (reconstruction_decoder.py:111-116: a
// column scaling, exact), and a copy of the sampling offset -- the decoder's per-call inputs in ONE launch (three torch launches
// in rounds 1-4: a multiply and two device-to-device copies at the head of every step)
__global__ void decoder_inputs_kernel(const float* v2s, const float* offset, int B, int nscales, float f0, float f1, float f2, float f3,
                                      float* layer_mats, float* offset_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float f[4] = {f0, f1, f2, f3};
  if (e < nscales * B * 16) {
    const int s = e / (B * 16), r = e - s * B * 16;
    layer_mats[e] = (r & 3) == 3 ? v2s[r] : v2s[r] * f[s];
  }
  if (e < B * 3) offset_out[e] = offset[e];
}

__global__ void gather_kernel(const float* src, const int32_t* idx, float* dst, int64_t n) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t i = idx[e];
    dst[e] = i >= 0 ? src[i] : 0.f;
  }
}
__global__ void scatter_kernel(const float* src, const int32_t* idx, float* dst, int64_t n, int accumulate) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t i = idx[e];
    if (i >= 0) { if (accumulate) dst[i] += src[e]; else dst[i] = src[e]; }
  }
}
__global__ void add_i64_kernel(int64_t* p, int n, int64_t v) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) p[e] += v;
}


// ---- tiled index copies (weight pack / gradient un-pack) -----------------------------------------
// The packed conv layouts are transposes of the reference layouts: a flat int32 index per element
// costs 4 B of index traffic per 4 B moved and makes one side of the copy touch a different 32-B sector
// per lane.  Here the packed side is cut into 8x8 tiles (rows x 8 consecutive columns); a tile whose
// index map is affine (src = base + r*sr + c*sc, true for every plain conv) is described by 6 ints + a
// 64-bit validity mask, other tiles (transposed-conv windows, stem, bias) point to 64 explicit indices.
// One wavefront per tile: both sides move in 8 full 32-B sectors.
template <bool REVERSE>
__global__ __launch_bounds__(256) void copy_tiles_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const int32_t* __restrict__ desc,
                                                         const unsigned long long* __restrict__ mask,
                                                         const int32_t* __restrict__ ex, int64_t ntiles) {
  const int lane = threadIdx.x & 63;
  const int r = lane >> 3, c = lane & 7;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t tIdx = wave0; tIdx < ntiles; tIdx += nwaves) {
    const int32_t* d = desc + tIdx * 6;
    const unsigned long long m = mask[tIdx];
    if (!((m >> lane) & 1ull)) continue;
    const int64_t tile_pos = (int64_t)d[0] + (int64_t)r * d[1] + c;                   // packed side
    const int64_t index = d[5] >= 0 ? (int64_t)ex[(int64_t)d[5] + lane] : (int64_t)d[2] + (int64_t)r * d[3] + (int64_t)c * d[4];
    if (REVERSE) dst[index] = src[tile_pos];     // un-pack: reference layout <- packed
    else dst[tile_pos] = src[index];             // pack: packed <- reference layout
  }
}

// ---- block copies through LDS (weight pack / gradient un-pack of the plain convolutions) ----------------
// The 8x8 tiles above move 32-byte sectors on both sides; a plain conv's pack is a 2-D transpose (forward layout:
// [n][c*T + t] -> [c*T + t][n]) or a batch of small ones (data-gradient layout: per n, [c][t] -> [T-1-t][c]), which
// an LDS tile turns into runs of 64+ floats on BOTH sides.  A block is G x A x B elements (g, a, b):
//   reference index = fbase + g*fg + a*fa + b          (b runs along the reference layout's contiguous axis)
//   packed position = pbase + g*pg + a*pa + b*pb       (pa == 1: a is the packed row's contiguous axis; else pb == 1)
// desc[t] = {A, B, G, fbase, fa, fg, pbase, pa, pb, pg, 0, 0}; G*A*(B|1) <= MAT_LDS floats.
constexpr int MAT_LDS = 8448;
// desc[t] = {A, B, G, fbase, fa, fg, pbase, pa, pb, pg, ceil(2^32/(A*B)), ceil(2^32/B), ceil(2^32/A), 0, 0, 0}
// (the three reciprocals turn the index splits into one multiply-high each; element counts stay below 2^16).
// Eight elements per thread are in flight at a time: one load per loop trip left every thread waiting out a full
// memory round trip per element (71 us for the 0.3 GB of a data-gradient pack; the 8x8 tiles took 100).
__device__ __forceinline__ int mat_div(int x, int d, unsigned magic) { return d == 1 ? x : (int)__umulhi((unsigned)x, magic); }
template <bool REVERSE>
__global__ __launch_bounds__(256) void copy_mats_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int32_t* __restrict__ desc, int64_t ntiles) {
  __shared__ float tile[MAT_LDS];
  constexpr int U = 8;
  for (int64_t tIdx = blockIdx.x; tIdx < ntiles; tIdx += gridDim.x) {
    const int32_t* d = desc + tIdx * 16;
    const int A = d[0], B = d[1], G = d[2];
    const int64_t fbase = d[3], fa = d[4], fg = d[5], pbase = d[6], pa = d[7], pb = d[8], pg = d[9];
    const unsigned mAB = (unsigned)d[10], mB = (unsigned)d[11], mA = (unsigned)d[12];
    const int Bp = B | 1, AB = A * B, n = G * AB;
    const bool a_fast = pa == 1;
    // the two walks over the block: b fastest (reference side; packed side too when pb == 1), a fastest (packed side)
    auto ref_side = [&](int e, int& l, int64_t& addr) {
      const int g = mat_div(e, AB, mAB), rem = e - g * AB, a = mat_div(rem, B, mB), b = rem - a * B;
      l = (g * A + a) * Bp + b; addr = fbase + g * fg + a * fa + b;
    };
    auto packed_side = [&](int e, int& l, int64_t& addr) {
      const int g = mat_div(e, AB, mAB), rem = e - g * AB;
      int a, b;
      if (a_fast) { b = mat_div(rem, A, mA); a = rem - b * A; } else { a = mat_div(rem, B, mB); b = rem - a * B; }
      l = (g * A + a) * Bp + b; addr = pbase + g * pg + a * pa + b * pb;
    };
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
      int l[U]; float val[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256;
        int64_t addr = 0;
        l[u] = -1;
        if (e < n) { if (REVERSE) packed_side(e, l[u], addr); else ref_side(e, l[u], addr); }
        val[u] = e < n ? src[addr] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (l[u] >= 0) tile[l[u]] = val[u];
    }
    __syncthreads();
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0 + u * 256;
        if (e >= n) break;
        int l; int64_t addr;
        if (REVERSE) ref_side(e, l, addr); else packed_side(e, l, addr);
        dst[addr] = tile[l];
      }
    }
    __syncthreads();
  }
}

// torch.optim.Adam (amsgrad=False, weight_decay=0), fp32, float4 per lane
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                   int64_t n, float lr, float b1, float b2, float eps,
                                                   float gs, float bc1, float bc2_sqrt) {
  const float step = lr / bc1;
  auto one = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= gs;
    mm = mm + (1.f - b1) * (gg - mm);
    vv = b2 * vv + (1.f - b2) * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float P[4], M[4], V[4], G[4];
    *reinterpret_cast<f32x4*>(P) = reinterpret_cast<f32x4*>(p)[i];
    *reinterpret_cast<f32x4*>(M) = reinterpret_cast<f32x4*>(m)[i];
    *reinterpret_cast<f32x4*>(V) = reinterpret_cast<f32x4*>(v)[i];
    *reinterpret_cast<f32x4*>(G) = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) one(P[k], G[k], M[k], V[k]);
    reinterpret_cast<f32x4*>(p)[i] = *reinterpret_cast<f32x4*>(P);
    reinterpret_cast<f32x4*>(m)[i] = *reinterpret_cast<f32x4*>(M);
    reinterpret_cast<f32x4*>(v)[i] = *reinterpret_cast<f32x4*>(V);
  }
  if (blockIdx.x == 0) {
    const int64_t i = n4 * 4 + threadIdx.x;
    if (i < n) one(p[i], g[i], m[i], v[i]);
  }
}

// The same step with its scalars in device memory (hyper = lr, beta1, beta2, eps, grad_scale, 1 - beta1^t,
// sqrt(1 - beta2^t)): the launch carries no per-step host value, so it can sit in a captured HIP graph that is
// replayed every step (crn_adam_set_hyper refreshes the seven floats with an ordinary launch before the replay).
__global__ __launch_bounds__(256) void adam_hyper_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t n4,
                                                         int64_t n, const float* __restrict__ hyper) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], gs = hyper[4], bc1 = hyper[5], bc2_sqrt = hyper[6];
  const float step = lr / bc1;
  auto one = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= gs;
    mm = mm + (1.f - b1) * (gg - mm);
    vv = b2 * vv + (1.f - b2) * gg * gg;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step * (mm / denom);
  };
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float P[4], M[4], V[4], G[4];
    *reinterpret_cast<f32x4*>(P) = reinterpret_cast<f32x4*>(p)[i];
    *reinterpret_cast<f32x4*>(M) = reinterpret_cast<f32x4*>(m)[i];
    *reinterpret_cast<f32x4*>(V) = reinterpret_cast<f32x4*>(v)[i];
    *reinterpret_cast<f32x4*>(G) = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) one(P[k], G[k], M[k], V[k]);
    reinterpret_cast<f32x4*>(p)[i] = *reinterpret_cast<f32x4*>(P);
    reinterpret_cast<f32x4*>(m)[i] = *reinterpret_cast<f32x4*>(M);
    reinterpret_cast<f32x4*>(v)[i] = *reinterpret_cast<f32x4*>(V);
  }
  if (blockIdx.x == 0) {
    const int64_t i = n4 * 4 + threadIdx.x;
    if (i < n) one(p[i], g[i], m[i], v[i]);
  }
}

__global__ void set_hyper_kernel(float* hyper, float a0, float a1, float a2, float a3, float a4, float a5, float a6) {
  if (threadIdx.x == 0) {
    hyper[0] = a0; hyper[1] = a1; hyper[2] = a2; hyper[3] = a3; hyper[4] = a4; hyper[5] = a5; hyper[6] = a6;
  }
}

inline unsigned nblk(int64_t n, int per = 256) { return (unsigned)std::max<int64_t>(1, (n + per - 1) / per); }

}  // namespace

// four pixels per thread (one 4-byte load, one 16-byte store), 32-bit index arithmetic: the scalar kernel above spends
// its 16 us on three 64-bit divisions per byte
__global__ __launch_bounds__(256) void preprocess4_kernel(const uint8_t* img, int HW4, float* out, int total4) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const int plane = e / HW4, s4 = e - plane * HW4;      // plane = b*3 + c
  const int b = plane / 3, c = plane - 3 * b;
  const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
  const uchar4 v = reinterpret_cast<const uchar4*>(img)[(b * 3 + (2 - c)) * HW4 + s4];
  reinterpret_cast<float4*>(out)[e] = make_float4((float)v.x + mean, (float)v.y + mean, (float)v.z + mean, (float)v.w + mean);
}

extern "C" int crn_preprocess_caffe(const uint8_t* img, int B, int H, int W, float* out, crnStream s) {
  CRN_ENTRY(s);
  const int64_t total = (int64_t)B * 3 * H * W;
  if (((int64_t)H * W) % 4 == 0 && total / 4 < (1 << 30) && (((uintptr_t)img) & 3) == 0 && (((uintptr_t)out) & 15) == 0) {
    hipLaunchKernelGGL(preprocess4_kernel, dim3(nblk(total / 4)), dim3(256), 0, (hipStream_t)s, img, (int)((int64_t)H * W / 4),
                       out, (int)(total / 4));
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  hipLaunchKernelGGL(preprocess_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)s, img, (int64_t)H * W, out, total);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_bn_relu_maxpool_fwd(const float* x, const float* scale, const float* shift, int B, int C,
                                       int H, int W, float* y, int32_t* argmax, crnStream s) {
  CRN_ENTRY(s);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * C * Ho * Wo;
  hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)s, x, scale, shift,
                     C, H, W, Ho, Wo, y, argmax, total);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_bn_relu_maxpool_bwd(const float* dy, const int32_t* argmax, int B, int C, int H, int W,
                                       float* dx_bn, crnStream s) {
  CRN_ENTRY(s);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(bn_relu_maxpool_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)s, dy, argmax, H, W,
                     Ho, Wo, dx_bn, total);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_relu_mean_fwd(const float* x_pre, int B, int C, int64_t S, int64_t sB, float* avg, crnStream s) {
  CRN_ENTRY(s);
  const int BC = B * C;
  hipLaunchKernelGGL(relu_mean_fwd_kernel, dim3(nblk((int64_t)BC * 64)), dim3(256), 0, (hipStream_t)s, x_pre, C, S,
                     sB, avg, BC);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_relu_mean_bwd(const float* x_pre, const float* davg, int B, int C, int64_t S, int64_t sB,
                                 float* dx, int64_t sB_dx, int accumulate, crnStream s) {
  CRN_ENTRY(s);
  const int64_t total = (int64_t)B * C * S;
  hipLaunchKernelGGL(relu_mean_bwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)s, x_pre, davg, C, S, sB,
                     dx, sB_dx, accumulate, total);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_linear_fwd(const float* x, const float* w, const float* bias, int B, int K, int N, float* y,
                              int ldy, crnStream s) {
  CRN_ENTRY(s);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(nblk((int64_t)B * N * 64)), dim3(256), 0, (hipStream_t)s, x, w, bias, B,
                     K, N, y, ldy);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_linear_bwd(const float* x, const float* w, const float* dy, int lddy, int B, int K, int N,
                              float* dx, float* dw, float* db, crnStream s) {
  CRN_ENTRY(s);
  if (dx) {
    hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(nblk((int64_t)B * K)), dim3(256), 0, (hipStream_t)s, w, dy, lddy, B,
                       K, N, dx);
    CRN_CHECK_LAUNCH();
  }
  if (dw) {
    hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(nblk((int64_t)N * K)), dim3(256), 0, (hipStream_t)s, x, dy, lddy, B,
                       K, N, dw, db);
    CRN_CHECK_LAUNCH();
  }
  return CRN_OK;
}

extern "C" int crn_fill_offset_channels(float* x, int B, int64_t sB, int64_t S, int c0, const float* offset,
                                        crnStream s) {
  CRN_ENTRY(s);
  const int64_t total = (int64_t)B * 3 * S;
  hipLaunchKernelGGL(fill_offset_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)s, x, sB, S, c0, offset, total);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_decoder_inputs(const float* v2s, const float* offset, int B, int nscales, const float* scales, float* layer_mats,
                                  float* offset_out, crnStream s) {
  CRN_ENTRY(s);
  if (!v2s || !offset || !scales || !layer_mats || !offset_out || B < 1 || nscales < 1 || nscales > 4) return CRN_EINVAL;
  float f[4] = {1.f, 1.f, 1.f, 1.f};
  for (int i = 0; i < nscales; ++i) f[i] = scales[i];          // (host array)
  hipLaunchKernelGGL(decoder_inputs_kernel, dim3(nblk((int64_t)nscales * B * 16)), dim3(256), 0, (hipStream_t)s, v2s, offset, B, nscales,
                     f[0], f[1], f[2], f[3], layer_mats, offset_out);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_gather_f32(const float* src, const int32_t* idx, float* dst, int64_t n, crnStream s) {
  CRN_ENTRY(s);
  hipLaunchKernelGGL(gather_kernel, dim3(std::min(nblk(n), 4096u)), dim3(256), 0, (hipStream_t)s, src, idx, dst, n);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_scatter_f32(const float* src, const int32_t* idx, float* dst, int64_t n, int accumulate,
                               crnStream s) {
  CRN_ENTRY(s);
  hipLaunchKernelGGL(scatter_kernel, dim3(std::min(nblk(n), 4096u)), dim3(256), 0, (hipStream_t)s, src, idx, dst, n,
                     accumulate);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_copy_tiles_f32(const float* src, float* dst, const int32_t* desc, const uint64_t* mask,
                                  const int32_t* explicit_idx, int64_t ntiles, int reverse, crnStream s) {
  CRN_ENTRY(s);
  if (!src || !dst || !desc || !mask || ntiles < 0) return CRN_EINVAL;
  if (ntiles == 0) return CRN_OK;
  // (grid-stride loops: the cap decides how much of the chip a pack / un-pack occupies while it runs BESIDE the step's latency-bound
  // launches on the other stream -- CRN_COPY_TILES_BLOCKS)
  static const int64_t kCap = getenv("CRN_COPY_TILES_BLOCKS") ? std::max(1, atoi(getenv("CRN_COPY_TILES_BLOCKS"))) : 16384;
  const unsigned blocks = (unsigned)std::min<int64_t>(crn_cdiv(ntiles, 4), kCap);
  if (reverse)
    hipLaunchKernelGGL(copy_tiles_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, dst, desc,
                       reinterpret_cast<const unsigned long long*>(mask), explicit_idx, ntiles);
  else
    hipLaunchKernelGGL(copy_tiles_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, dst, desc,
                       reinterpret_cast<const unsigned long long*>(mask), explicit_idx, ntiles);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_copy_mats_f32(const float* src, float* dst, const int32_t* desc, int64_t ntiles, int reverse, crnStream s) {
  CRN_ENTRY(s);
  if (!src || !dst || !desc || ntiles < 0) return CRN_EINVAL;
  if (ntiles == 0) return CRN_OK;
  static const int64_t kCap = getenv("CRN_COPY_MATS_BLOCKS") ? std::max(1, atoi(getenv("CRN_COPY_MATS_BLOCKS"))) : 8192;
  const unsigned blocks = (unsigned)std::min<int64_t>(ntiles, kCap);
  if (reverse)
    hipLaunchKernelGGL(copy_mats_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, dst, desc, ntiles);
  else
    hipLaunchKernelGGL(copy_mats_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)s, src, dst, desc, ntiles);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_zero_f32(float* p, int64_t n, crnStream s) {
  CRN_ENTRY(s);
  CRN_HIP(hipMemsetAsync(p, 0, (size_t)n * 4, (hipStream_t)s));
  return CRN_OK;
}

extern "C" int crn_add_i64(int64_t* p, int n, int64_t v, crnStream s) {
  CRN_ENTRY(s);
  hipLaunchKernelGGL(add_i64_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)s, p, n, v);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, float grad_scale, int step,
                             crnStream s) {
  CRN_ENTRY(s);
  if (step < 1 || n < 1) return CRN_EINVAL;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return CRN_EINVAL;
  const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(std::min(nblk(n4), 8192u)), dim3(256), 0, (hipStream_t)s, param, grad, exp_avg,
                     exp_avg_sq, n4, n, lr, beta1, beta2, eps, grad_scale, (float)bc1, (float)std::sqrt(bc2));
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_adam_set_hyper(float* hyper, float lr, float beta1, float beta2, float eps, float grad_scale,
                                  int step, crnStream s) {
  CRN_ENTRY(s);
  if (!hyper || step < 1) return CRN_EINVAL;
  const double bc1 = 1.0 - std::pow((double)beta1, step), bc2 = 1.0 - std::pow((double)beta2, step);
  hipLaunchKernelGGL(set_hyper_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, hyper, lr, beta1, beta2, eps, grad_scale,
                     (float)bc1, (float)std::sqrt(bc2));
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_adam_step_hyper(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                                   const float* hyper, crnStream s) {
  CRN_ENTRY(s);
  if (n < 1 || !hyper) return CRN_EINVAL;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return CRN_EINVAL;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(std::min(nblk(n4), 8192u)), dim3(256), 0, (hipStream_t)s, param, grad,
                     exp_avg, exp_avg_sq, n4, n, hyper);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

// Stride-2 1x1 convolutions of the ResNet downscale blocks (resnet50.py:94-97: stride on the first 1x1 and on the
// shortcut): the sub-sampled input is compacted once, y[b,c,i,j] = x[b,c,2i,2j], so that both convolutions (and
// their weight gradients) run on a plain tensor with the pointwise kernel instead of on a strided view; the
// data gradient goes the other way, dx[b,c,2i,2j] = dy[b,c,i,j], zeros elsewhere (which also replaces the memset).
namespace {
__global__ __launch_bounds__(256) void stride2_gather_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int h, int w, int64_t planes) {
  // one thread = 2 output elements (one float4 of input: elements 0 and 2)
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int w2 = w >> 1;
  const int64_t n = planes * h * w2;
  if (t >= n) return;
  const int j2 = (int)(t % w2);
  const int64_t r = t / w2;
  const int i = (int)(r % h);
  const int64_t p = r / h;
  const f32x4 v = *reinterpret_cast<const f32x4*>(x + (p * (2 * h) + 2 * i) * (int64_t)(2 * w) + 4 * j2);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  *reinterpret_cast<f32x2*>(y + (p * h + i) * (int64_t)w + 2 * j2) = (f32x2){v[0], v[2]};
}
__global__ __launch_bounds__(256) void stride2_scatter_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                              int h, int w, int64_t planes) {
  // one thread = one float4 of the output row (2i or 2i+1): 2 input elements on even rows, zeros on odd rows
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int w2 = w >> 1;
  const int64_t n = planes * (2 * h) * w2;
  if (t >= n) return;
  const int j2 = (int)(t % w2);
  const int64_t r = t / w2;
  const int io = (int)(r % (2 * h));
  const int64_t p = r / (2 * h);
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
  if ((io & 1) == 0) {
    const float* s = dy + (p * h + (io >> 1)) * (int64_t)w + 2 * j2;
    v[0] = s[0]; v[2] = s[1];
  }
  *reinterpret_cast<f32x4*>(dx + (p * (2 * h) + io) * (int64_t)(2 * w) + 4 * j2) = v;
}
// any size (odd output widths, odd input extents: images other than 256 x 256): one thread per element
__global__ __launch_bounds__(256) void stride2_gather_any_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w,
                                                                 int hin, int win, int64_t planes) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= planes * h * w) return;
  const int j = (int)(t % w);
  const int64_t r = t / w;
  const int i = (int)(r % h);
  const int64_t p = r / h;
  y[t] = x[(p * hin + 2 * i) * (int64_t)win + 2 * j];
}
__global__ __launch_bounds__(256) void stride2_scatter_any_kernel(const float* __restrict__ dy, float* __restrict__ dx, int h, int w,
                                                                  int hin, int win, int64_t planes) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= planes * hin * win) return;
  const int jo = (int)(t % win);
  const int64_t r = t / win;
  const int io = (int)(r % hin);
  const int64_t p = r / hin;
  dx[t] = ((io | jo) & 1) ? 0.f : dy[(p * h + (io >> 1)) * (int64_t)w + (jo >> 1)];
}
}  // namespace

extern "C" int crn_stride2_gather(const float* x, float* y, int B, int C, int h, int w, int hin, int win, crnStream s) {
  CRN_ENTRY(s);
  if (!x || !y || B < 1 || C < 1 || h < 1 || w < 1 || h != (hin + 1) / 2 || w != (win + 1) / 2) return CRN_EINVAL;
  const int64_t planes = (int64_t)B * C;
  if (hin == 2 * h && win == 2 * w && !(w & 1) && !(((uintptr_t)x) & 15) && !(((uintptr_t)y) & 7)) {
    hipLaunchKernelGGL(stride2_gather_kernel, dim3(nblk(planes * h * (w >> 1))), dim3(256), 0, (hipStream_t)s, x, y, h, w, planes);
  } else {
    hipLaunchKernelGGL(stride2_gather_any_kernel, dim3(nblk(planes * h * w)), dim3(256), 0, (hipStream_t)s, x, y, h, w, hin, win, planes);
  }
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_stride2_scatter(const float* dy, float* dx, int B, int C, int h, int w, int hin, int win, crnStream s) {
  CRN_ENTRY(s);
  if (!dy || !dx || B < 1 || C < 1 || h < 1 || w < 1 || h != (hin + 1) / 2 || w != (win + 1) / 2) return CRN_EINVAL;
  const int64_t planes = (int64_t)B * C;
  if (hin == 2 * h && win == 2 * w && !(w & 1) && !(((uintptr_t)dx) & 15)) {
    hipLaunchKernelGGL(stride2_scatter_kernel, dim3(nblk(planes * (2 * h) * (w >> 1))), dim3(256), 0, (hipStream_t)s, dy, dx, h, w, planes);
  } else {
    hipLaunchKernelGGL(stride2_scatter_any_kernel, dim3(nblk(planes * hin * win)), dim3(256), 0, (hipStream_t)s, dy, dx, h, w, hin, win, planes);
  }
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}



static int g_crn_det = -1;
bool crn_deterministic() {
  if (g_crn_det < 0) { const char* e = getenv("CRN_DETERMINISTIC"); g_crn_det = (e && atoi(e) != 0) ? 1 : 0; }
  return g_crn_det != 0;
}
extern "C" int crn_set_deterministic(int on) { g_crn_det = on ? 1 : 0; return CRN_OK; }

extern "C" const char* crn_version(void) { return "corenet_hip 0.1 (gfx950)"; }
