// BatchRenorm statistics / backward and the fused per-channel elementwise
// kernels of the encoder/decoder blocks.  All HBM-bound: one coalesced float4
// pass per tensor, fp64 accumulation of the per-channel sums.
// Reference: model/batch_renorm.py:33-62, model/resnet50.py:72-82,109-115.
#include "crn_common.h"
#include <cstdlib>
#include <algorithm>

namespace {

constexpr int kThreads = 256;

struct Slice { int64_t s0, s1; };
__device__ __forceinline__ Slice block_slice(int64_t S) {
  int64_t chunk = (S + gridDim.x - 1) / gridDim.x;
  chunk = (chunk + 3) & ~(int64_t)3;
  Slice r;
  r.s0 = (int64_t)blockIdx.x * chunk;
  r.s1 = min(S, r.s0 + chunk);
  return r;
}

// Applies f(s, vec4) over a contiguous slice; VEC needs 16-B aligned rows.
template <bool VEC, typename F4, typename F1>
__device__ __forceinline__ void for_slice(Slice sl, F4 f4, F1 f1) {
  if (VEC) {
#pragma unroll 4
    for (int64_t s = sl.s0 + (int64_t)threadIdx.x * 4; s + 3 < sl.s1; s += (int64_t)kThreads * 4) f4(s);
    const int64_t nfull = (sl.s1 > sl.s0) ? ((sl.s1 - sl.s0) & ~(int64_t)3) : 0;
    for (int64_t s = sl.s0 + nfull + threadIdx.x; s < sl.s1; s += kThreads) f1(s);
  } else {
    for (int64_t s = sl.s0 + threadIdx.x; s < sl.s1; s += kThreads) f1(s);
  }
}

// ---- statistics -------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_partial_kernel(const float* x, int64_t S, int64_t sB,
                                                               int pre_relu, double* ws) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.y, b = blockIdx.z;
  const float* p = x + (int64_t)b * sB + (int64_t)c * S;
  double s1 = 0.0, s2 = 0.0;
  auto one = [&](float v) {
    if (pre_relu) v = fmaxf(v, 0.f);
    s1 += (double)v;
    s2 += (double)v * (double)v;
  };
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + s);
                                  one(v.x); one(v.y); one(v.z); one(v.w); },
                 [&](int64_t s) { one(p[s]); });
  const double t1 = crn_block_sum(s1, red);
  const double t2 = crn_block_sum(s2, red);
  if (threadIdx.x == 0) {
    const int nparts = gridDim.x * gridDim.z;
    double* o = ws + ((int64_t)c * nparts + (int64_t)b * gridDim.x + blockIdx.x) * 2;
    o[0] = t1; o[1] = t2;
  }
}

// One channel's statistics -> scale/shift, saved values, running statistics (batch_renorm.py:33-62).
// The five per-channel inputs of bn_finalize_channel, loaded at the START of an owner kernel (they do not depend on the
// statistics): their round trip to memory then overlaps the one of x instead of following the reduction.
struct BnChannelIn { float g, bt, rmean, rvar, nt; };
__device__ __forceinline__ BnChannelIn bn_channel_in(int c, const float* gamma, const float* beta,
                                                     const float* running_mean, const float* running_var,
                                                     const int64_t* nbt) {
  return BnChannelIn{gamma[c], beta[c], running_mean[c], running_var[c], (float)nbt[0]};
}
__device__ __forceinline__ void bn_finalize_channel_in(int c, int C, double s1, double s2, double count,
                                                       const BnChannelIn& in, float* running_mean, float* running_var,
                                                       float eps, float momentum, float* scale, float* shift,
                                                       float* saved) {
  const float g = in.g, bt = in.bt;
  const double mean_d = s1 / count;
  double var_d = s2 / count - mean_d * mean_d;
  if (var_d < 0.0) var_d = 0.0;
  const float b_mean = (float)mean_d, b_var = (float)var_d;
  const float b_std = sqrtf(b_var + eps);
  const float run_std = sqrtf(in.rvar + eps);
  const float nt = in.nt;
  const float d_max = fminf(fmaxf(5.0f * (nt - 5000.f) / (25000.f - 5000.f), 0.f), 5.f);
  const float r_max = 1.0f + fminf(fmaxf(2.0f * (nt - 5000.f) / (40000.f - 5000.f), 0.f), 2.f);
  float r = b_std / run_std;
  r = fminf(fmaxf(r, 1.0f / r_max), r_max);
  float d = (b_mean - in.rmean) / run_std;
  d = fminf(fmaxf(d, -d_max), d_max);
  const float rstd = 1.0f / b_std;
  scale[c] = g * r * rstd;
  shift[c] = bt + g * (d - b_mean * r * rstd);
  saved[c] = b_mean; saved[C + c] = rstd; saved[2 * C + c] = r; saved[3 * C + c] = d;
  const float unbiased = b_var * (float)C / (float)(C - 1);
  running_var[c] = in.rvar + momentum * (unbiased - in.rvar);
  running_mean[c] = in.rmean + momentum * (b_mean - in.rmean);
}

__device__ __forceinline__ void bn_finalize_channel(int c, int C, double s1, double s2, double count,
                                                    const float* gamma, const float* beta, float* running_mean,
                                                    float* running_var, const int64_t* nbt, float eps,
                                                    float momentum, float* scale, float* shift, float* saved) {
  const float g = gamma[c], bt = beta[c];
  const double mean_d = s1 / count;
  double var_d = s2 / count - mean_d * mean_d;
  if (var_d < 0.0) var_d = 0.0;
  const float b_mean = (float)mean_d, b_var = (float)var_d;
  const float b_std = sqrtf(b_var + eps);
  const float run_std = sqrtf(running_var[c] + eps);
  // batch_renorm.py:41-42: schedules from num_batches_tracked (before the increment)
  const float nt = (float)nbt[0];
  const float d_max = fminf(fmaxf(5.0f * (nt - 5000.f) / (25000.f - 5000.f), 0.f), 5.f);
  const float r_max = 1.0f + fminf(fmaxf(2.0f * (nt - 5000.f) / (40000.f - 5000.f), 0.f), 2.f);
  float r = b_std / run_std;
  r = fminf(fmaxf(r, 1.0f / r_max), r_max);
  float d = (b_mean - running_mean[c]) / run_std;
  d = fminf(fmaxf(d, -d_max), d_max);
  const float rstd = 1.0f / b_std;
  scale[c] = g * r * rstd;
  shift[c] = bt + g * (d - b_mean * r * rstd);
  saved[c] = b_mean; saved[C + c] = rstd; saved[2 * C + c] = r; saved[3 * C + c] = d;
  // batch_renorm.py:54-56 (the "unbiased" factor uses C = channel count)
  const float unbiased = b_var * (float)C / (float)(C - 1);
  running_var[c] += momentum * (unbiased - running_var[c]);
  running_mean[c] += momentum * (b_mean - running_mean[c]);
}

__global__ void bn_finalize_kernel(const double* ws, int nparts, int C, double count,
                                   const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, const int64_t* nbt, float eps, float momentum,
                                   int training, float* scale, float* shift, float* saved) {
  // one wave per channel: its lanes add the channel's partial sums up together (one thread per channel walked up to
  // 4096 of them one memory latency after the other: 9 us for a 100-channel launch)
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  if (!training) {          // batch_renorm.py:59: (x - running_mean) / running_std
    if (lane) return;
    const float g = gamma[c], bt = beta[c];
    const float rstd = 1.0f / sqrtf(running_var[c] + eps);
    scale[c] = g * rstd;
    shift[c] = bt - g * running_mean[c] * rstd;
    return;
  }
  const BnChannelIn cin = bn_channel_in(c, gamma, beta, running_mean, running_var, nbt);
  double s1 = 0.0, s2 = 0.0;
  for (int i = lane; i < nparts; i += 64) { s1 += ws[((int64_t)c * nparts + i) * 2]; s2 += ws[((int64_t)c * nparts + i) * 2 + 1]; }
  s1 = crn_wave_sum(s1); s2 = crn_wave_sum(s2);
  if (lane == 0)
    bn_finalize_channel_in(c, C, s1, s2, count, cin, running_mean, running_var, eps, momentum, scale, shift, saved);
}

// The same for MANY parts per channel (the partial sums a convolution's workgroups leave: 2048 for decoder stage_6.c1): one
// workgroup per channel, every thread has its share of the loads in flight at once (a wave's 64 lanes walked 32 of them one
// after the other: 12.7 us for 16 channels)
__global__ __launch_bounds__(kThreads) void bn_finalize_wide_kernel(const double* ws, int nparts, int C, double count,
                                                                    const float* gamma, const float* beta, float* running_mean,
                                                                    float* running_var, const int64_t* nbt, float eps, float momentum,
                                                                    float* scale, float* shift, float* saved) {
  __shared__ double red[2 * (kThreads / 64)];
  const int c = blockIdx.x;
  const BnChannelIn cin = bn_channel_in(c, gamma, beta, running_mean, running_var, nbt);
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const f64x2* p = reinterpret_cast<const f64x2*>(ws) + (int64_t)c * nparts;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
  for (int i = threadIdx.x; i < nparts; i += kThreads) { const f64x2 v = p[i]; s1 += v.x; s2 += v.y; }
  double t1, t2;
  crn_block_sum2(s1, s2, red, t1, t2);
  if (threadIdx.x == 0)
    bn_finalize_channel_in(c, C, t1, t2, count, cin, running_mean, running_var, eps, momentum, scale, shift, saved);
}

// Eval mode: scale/shift of EVERY BatchRenorm of the model from the running statistics in one launch
// (batch_renorm.py:59); table rows = (gamma, beta offsets in the parameter slab; running_mean, running_var
// offsets in the buffer slab; output offset in the scale/shift slabs).  Same arithmetic as the eval branch of
// bn_finalize_kernel.
__global__ void bn_eval_affine_kernel(const float* params, const float* buffers, const int32_t* table, int n,
                                      float eps, float* scale, float* shift) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int32_t* r = table + (int64_t)j * 5;
  const float g = params[r[0]], bt = params[r[1]];
  const float rstd = 1.0f / sqrtf(buffers[r[3]] + eps);
  scale[r[4]] = g * rstd;
  shift[r[4]] = bt - g * buffers[r[2]] * rstd;
}

// Channel-owner form for many small channels (the encoder from stage 3 on, the coarse decoder stages): one
// workgroup reduces all of (B, S) of its channel and finalizes it -- one launch instead of two, no workspace.
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_owner_stats_kernel(
    const float* x, int B, int C, int64_t S, int64_t sB, int pre_relu, const float* gamma, const float* beta,
    float* running_mean, float* running_var, const int64_t* nbt, float eps, float momentum, float* scale,
    float* shift, float* saved) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  auto one = [&](float v) {
    if (pre_relu) v = fmaxf(v, 0.f);
    s1 += (double)v;
    s2 += (double)v * (double)v;
  };
  Slice all; all.s0 = 0; all.s1 = S;
  for (int b = 0; b < B; ++b) {
    const float* p = x + (int64_t)b * sB + (int64_t)c * S;
    for_slice<VEC>(all,
                   [&](int64_t s) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + s);
                                    one(v.x); one(v.y); one(v.z); one(v.w); },
                   [&](int64_t s) { one(p[s]); });
  }
  const double t1 = crn_block_sum(s1, red);
  const double t2 = crn_block_sum(s2, red);
  if (threadIdx.x == 0)
    bn_finalize_channel(c, C, t1, t2, (double)B * (double)S, gamma, beta, running_mean, running_var, nbt, eps,
                        momentum, scale, shift, saved);
}

// Register-resident owner kernels: when all of (B, S) of a channel fits in NV float4 per thread (B*S <= 1024*NV), every
// thread issues its NV loads at once (the loops above walk them one latency after the other: these launches are
// latency bound) and the backward kernel makes its second pass over registers instead of re-reading x and dy.
// part != nullptr: x is the output of a split-K convolution whose `splits` partial sums are still in the scratch
// ([split][b][c][pos], dense): they are added up here (split 0 first, the order of the reduction kernel) and the
// sum is stored to x on the way -- one launch instead of reduction + statistics (crn_splitk_defer).
// TAIL: the block tail of a ResNet bottleneck in the same launch -- the channel is in registers and its scale / shift are
// known to this workgroup the moment the statistics are finalized, so y = relu(x*scale + shift + residual) (the
// arithmetic of affine_add_relu_kernel, expression for expression) is written from here: 16 launches fewer per forward
// and one read of x instead of two.
// y2: the stride-2 compaction of y (y2[b,c,i,j] = y[b,c,2i,2j], crn_stride2_gather) for the down-sampling block that reads this
// block's output next (resnet50.py:94-97): written from the same registers, one launch less on the forward chain.  W = row width of
// the channel plane (a multiple of 4: a float4 never straddles rows), W2 = ceil(W / 2), H2 = ceil(H / 2).
struct BnTail {
  const float* r; const float* rscale; const float* rshift; int64_t sBr;
  float* y_pre; int64_t sBpre; float* y; int64_t sBy; int relu;
  float* y2; int64_t sBy2; int W, W2, H2;
};
template <int NV, bool TAIL = false>
__global__ __launch_bounds__(kThreads) void bn_owner_stats_reg_kernel(
    float* x, int B, int C, int S4, int64_t sB, int pre_relu, const float* gamma, const float* beta,
    float* running_mean, float* running_var, const int64_t* nbt, float eps, float momentum, float* scale,
    float* shift, float* saved, const float* part, int splits, BnTail tail = BnTail{}) {
  __shared__ double red[2 * (kThreads / 64)];
  __shared__ float tail_affine[2];
  crn_kernargs_now(x, B, C, S4, sB, pre_relu, gamma, beta, running_mean, running_var, nbt, eps, momentum, scale, shift,
                   saved, part, splits);
  if (TAIL) crn_kernargs_now(tail.r, tail.rscale, tail.rshift, tail.sBr, tail.y_pre, tail.sBpre, tail.y, tail.sBy, tail.relu);
  const int c = blockIdx.x, total4 = B * S4;
  const BnChannelIn cin = bn_channel_in(c, gamma, beta, running_mean, running_var, nbt);
  f32x4 v[NV];
  f32x4 rv[TAIL ? NV : 1];
  float rsc = 1.f, rsh = 0.f;
  if (TAIL && tail.r) { rsc = tail.rscale ? tail.rscale[c] : 1.f; rsh = tail.rshift ? tail.rshift[c] : 0.f; }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int b = e / S4, s4 = e - b * S4;
    f32x4* xp = reinterpret_cast<f32x4*>(x + (int64_t)b * sB + ((int64_t)c * S4 + s4) * 4);
    if (TAIL)      // the residual rides on the same round trip
      rv[k] = (tail.r && e < total4) ? *reinterpret_cast<const f32x4*>(tail.r + (int64_t)b * tail.sBr + ((int64_t)c * S4 + s4) * 4)
                                     : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (part == nullptr) {
      v[k] = e < total4 ? *xp : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
      f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (e < total4) {
        const f32x4* pp = reinterpret_cast<const f32x4*>(part) + ((int64_t)b * C + c) * S4 + s4;
        const int64_t slab4 = (int64_t)B * C * S4;
        for (int sp = 0; sp < splits; ++sp) sum += pp[sp * slab4];
        *xp = sum;
      }
      v[k] = sum;
    }
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = v[k][i];
      if (pre_relu) t = fmaxf(t, 0.f);
      s1 += (double)t;
      s2 += (double)t * (double)t;
    }
  double t1, t2;
  crn_block_sum2(s1, s2, red, t1, t2);
  if (threadIdx.x == 0) {
    bn_finalize_channel_in(c, C, t1, t2, (double)B * (double)S4 * 4.0, cin, running_mean, running_var, eps, momentum,
                           scale, shift, saved);
    if (TAIL) { tail_affine[0] = scale[c]; tail_affine[1] = shift[c]; }     // (the fp32 values the un-fused tail would load)
  }
  if (TAIL) {
    __syncthreads();
    const float sc = tail_affine[0], sh = tail_affine[1];
    const bool has_r = tail.r != nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int e = threadIdx.x + k * kThreads;
      if (e < total4) {
        const int b = e / S4, s4 = e - b * S4;
        const int64_t o = ((int64_t)c * S4 + s4) * 4;
        f32x4 q;
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = v[k][i] * sc + sh + (has_r ? rv[k][i] * rsc + rsh : 0.f);
        if (tail.y_pre) *reinterpret_cast<f32x4*>(tail.y_pre + (int64_t)b * tail.sBpre + o) = q;
        if (tail.y) {
          if (tail.relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = fmaxf(q[i], 0.f);
          }
          *reinterpret_cast<f32x4*>(tail.y + (int64_t)b * tail.sBy + o) = q;
          if (tail.y2) {
            const int s = s4 * 4, row = s / tail.W, col = s - row * tail.W;
            if (!(row & 1)) {
              typedef float f32x2 __attribute__((ext_vector_type(2)));
              float* d2 = tail.y2 + (int64_t)b * tail.sBy2 + ((int64_t)c * tail.H2 + (row >> 1)) * tail.W2 + (col >> 1);
              *reinterpret_cast<f32x2*>(d2) = (f32x2){q[0], q[2]};
            }
          }
        }
      }
    }
  }
}

// part != nullptr: dy is the output of a split-K convolution (a data gradient) still in `splits` partial sums in the
// scratch: added up while they are loaded; dy itself is NOT written (its only reader is this kernel).
// HEAD: the backward of a bottleneck's tail in the same launch -- dy is not read but formed,
//   dy = (act > 0 ? g : 0) + g2      (the arithmetic of relu_bwd_add_kernel; g2 may be absent),
// and stored to `dy` on the way (the shortcut's norm and the block's input gradient read it later): 15 launches fewer
// per backward.
// gc != nullptr: g is given in its compact form -- it is the data gradient of a stride-2 1x1 convolution, non-zero only at even
// (row, column) (crn_stride2_scatter(gc) would expand it: g[2i][2j] = gc[i][j], zeros elsewhere): read from there, the expanded
// tensor is never written.  W = row width of the full plane (a multiple of 4), W2 = ceil(W / 2), H2 = ceil(H / 2).
struct BnHead { const float* g; int64_t sBg; const float* act; int64_t sBact; const float* g2; int64_t sBg2;
                const float* gc; int64_t sBgc; int W, W2, H2; };
template <int NV, bool HEAD = false>
__global__ __launch_bounds__(kThreads) void bn_owner_bwd_reg_kernel(
    const float* x, int64_t sBx, const float* dy, int64_t sBdy, int B, int S4, int C, int pre_relu,
    int post_relu, const float* gamma, const float* scale, const float* shift, const float* saved, float* dx,
    int64_t sBdx, float* dgamma, float* dbeta, int accumulate, float* dsum, int ndsum, const float* part, int splits,
    BnHead head = BnHead{}) {
  __shared__ double red[2 * (kThreads / 64)];
  __shared__ float sm[2];
  __shared__ float redf[kThreads / 64];
  crn_kernargs_now(x, sBx, dy, sBdy, B, S4, C, pre_relu, post_relu, gamma, scale, shift, saved, dx, sBdx, dgamma, dbeta,
                   accumulate, dsum, ndsum, part, splits);
  if (HEAD) crn_kernargs_now(head.g, head.sBg, head.act, head.sBact, head.g2, head.sBg2, head.gc, head.sBgc, head.W, head.W2, head.H2);
  const int c = blockIdx.x, total4 = B * S4;
  // every per-channel scalar of the kernel is loaded here, next to x and dy: one round trip to memory, not three
  const float sc = scale[c], sh = shift[c], mu = saved[c], rstd = saved[C + c];
  const float rr = saved[2 * C + c], dd = saved[3 * C + c], gam = gamma[c];
  float dg0 = 0.f, db0 = 0.f;
  if (accumulate && threadIdx.x == 0) { dg0 = dgamma[c]; db0 = dbeta[c]; }
  f32x4 xv[NV], gv[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int b = e / S4, s4 = e - b * S4;
    const int64_t o = ((int64_t)c * S4 + s4) * 4;
    const bool ok = e < total4;
    xv[k] = ok ? *reinterpret_cast<const f32x4*>(x + (int64_t)b * sBx + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (HEAD) {
      const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 g = z;
      if (ok && head.gc) {
        const int s = s4 * 4, row = s / head.W, col = s - row * head.W;
        if (!(row & 1)) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 v = *reinterpret_cast<const f32x2*>(head.gc + (int64_t)b * head.sBgc + ((int64_t)c * head.H2 + (row >> 1)) * head.W2 + (col >> 1));
          g[0] = v[0]; g[2] = v[1];
        }
      } else if (ok) {
        g = *reinterpret_cast<const f32x4*>(head.g + (int64_t)b * head.sBg + o);
      }
      const f32x4 a = ok ? *reinterpret_cast<const f32x4*>(head.act + (int64_t)b * head.sBact + o) : z;
      const f32x4 g2 = (ok && head.g2) ? *reinterpret_cast<const f32x4*>(head.g2 + (int64_t)b * head.sBg2 + o) : z;
      f32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = (a[i] > 0.f ? g[i] : 0.f) + g2[i];
      if (ok) *reinterpret_cast<f32x4*>(const_cast<float*>(dy) + (int64_t)b * sBdy + o) = d;
      gv[k] = d;
    } else if (part == nullptr) {
      gv[k] = ok ? *reinterpret_cast<const f32x4*>(dy + (int64_t)b * sBdy + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
    } else {
      f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const f32x4* pp = reinterpret_cast<const f32x4*>(part) + ((int64_t)b * C + c) * S4 + s4;
        const int64_t slab4 = (int64_t)B * C * S4;
        for (int sp = 0; sp < splits; ++sp) sum += pp[sp * slab4];
      }
      gv[k] = sum;
    }
  }
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = xv[k][i], g = gv[k][i];
      if (pre_relu) a = fmaxf(a, 0.f);
      if (post_relu && !(a * sc + sh > 0.f)) g = 0.f;
      s1 += (double)g;
      s2 += (double)g * (double)((a - mu) * rstd);
    }
  double t1, t2;
  crn_block_sum2(s1, s2, red, t1, t2);
  if (threadIdx.x == 0) {
    const double count = (double)B * (double)S4 * 4.0;
    sm[0] = (float)(t1 / count); sm[1] = (float)(t2 / count);
    const float dg = (float)(rr * t2 + dd * t1), db = (float)t1;
    dgamma[c] = dg0 + dg; dbeta[c] = db0 + db;
  }
  __syncthreads();
  const float mg = sm[0], mgx = sm[1];
  const float kf = gam * rr * rstd;
  float lsum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int e = threadIdx.x + k * kThreads;
    if (e < total4) {
      const int b = e / S4, s4 = e - b * S4;
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xr = xv[k][i];
        const float a = pre_relu ? fmaxf(xr, 0.f) : xr;
        float g = gv[k][i];
        if (post_relu && !(a * sc + sh > 0.f)) g = 0.f;
        float r = kf * (g - mg - (a - mu) * rstd * mgx);
        if (pre_relu && !(xr > 0.f)) r = 0.f;
        lsum += r;
        o[i] = r;
      }
      *reinterpret_cast<f32x4*>(dx + (int64_t)b * sBdx + ((int64_t)c * S4 + s4) * 4) = o;
    }
  }
  if (dsum && c < ndsum) {       // bias gradient of the convolution that produced x: sum of dx
    const float w = crn_wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tsum = 0.f;
      for (int i = 0; i < kThreads / 64; ++i) tsum += redf[i];
      dsum[c] = tsum;
    }
  }
}

// ---- backward ----------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_partial_kernel(
    const float* x, int64_t sBx, const float* dy, int64_t sBdy, int64_t S, int C, int pre_relu,
    int post_relu, const float* scale, const float* shift, const float* saved, double* ws, float* dsum,
    int ndsum) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.y, b = blockIdx.z;
  // the apply kernel (next launch on this stream) accumulates sum(dx) here with atomics
  if (dsum && c < ndsum && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) dsum[c] = 0.f;
  const float* px = x + (int64_t)b * sBx + (int64_t)c * S;
  const float* pg = dy + (int64_t)b * sBdy + (int64_t)c * S;
  const float sc = scale[c], sh = shift[c], mu = saved[c], rstd = saved[C + c];
  double s1 = 0.0, s2 = 0.0;
  auto one = [&](float xv, float gv) {
    if (pre_relu) xv = fmaxf(xv, 0.f);
    if (post_relu && !(xv * sc + sh > 0.f)) gv = 0.f;
    s1 += (double)gv;
    s2 += (double)gv * (double)((xv - mu) * rstd);
  };
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) { const f32x4 a = *reinterpret_cast<const f32x4*>(px + s);
                                  const f32x4 g = *reinterpret_cast<const f32x4*>(pg + s);
                                  one(a.x, g.x); one(a.y, g.y); one(a.z, g.z); one(a.w, g.w); },
                 [&](int64_t s) { one(px[s], pg[s]); });
  const double t1 = crn_block_sum(s1, red);
  const double t2 = crn_block_sum(s2, red);
  if (threadIdx.x == 0) {
    const int nparts = gridDim.x * gridDim.z;
    double* o = ws + ((int64_t)c * nparts + (int64_t)b * gridDim.x + blockIdx.x) * 2;
    o[0] = t1; o[1] = t2;
  }
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    const float* x, int64_t sBx, const float* dy, int64_t sBdy, int64_t S, int C, int pre_relu,
    int post_relu, const float* gamma, const float* scale, const float* shift, const float* saved,
    const double* ws, int nparts, double count, float* dx, int64_t sBdx, float* dgamma,
    float* dbeta, int accumulate, float* dsum, int ndsum) {
  __shared__ float sm[2];
  __shared__ float red[kThreads / 64];
  __shared__ double redd[2 * (kThreads / 64)];
  const int c = blockIdx.y, b = blockIdx.z;
  // the partial sums of the channel are added up by the whole workgroup (one thread walking up to 4096 of them kept
  // the other 255 waiting at the barrier for several memory latencies), and the per-channel scalars used after the
  // barrier are loaded before it
  const float sc = scale[c], sh = shift[c], mu = saved[c], rstd = saved[C + c];
  const float k = gamma[c] * saved[2 * C + c] * rstd;
  double p1 = 0.0, p2 = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) { p1 += ws[((int64_t)c * nparts + i) * 2]; p2 += ws[((int64_t)c * nparts + i) * 2 + 1]; }
  double s1, s2;
  crn_block_sum2(p1, p2, redd, s1, s2);
  if (threadIdx.x == 0) {
    sm[0] = (float)(s1 / count); sm[1] = (float)(s2 / count);
    if (blockIdx.x == 0 && b == 0) {
      const float r = saved[2 * C + c], d = saved[3 * C + c];
      const float dg = (float)(r * s2 + d * s1), db = (float)s1;
      if (accumulate) { dgamma[c] += dg; dbeta[c] += db; } else { dgamma[c] = dg; dbeta[c] = db; }
    }
  }
  __syncthreads();
  const float mg = sm[0], mgx = sm[1];
  const float* px = x + (int64_t)b * sBx + (int64_t)c * S;
  const float* pg = dy + (int64_t)b * sBdy + (int64_t)c * S;
  float* po = dx + (int64_t)b * sBdx + (int64_t)c * S;
  float lsum = 0.f;
  auto one = [&](float xr, float gv) -> float {
    const float xv = pre_relu ? fmaxf(xr, 0.f) : xr;
    if (post_relu && !(xv * sc + sh > 0.f)) gv = 0.f;
    float o = k * (gv - mg - (xv - mu) * rstd * mgx);
    if (pre_relu && !(xr > 0.f)) o = 0.f;
    lsum += o;
    return o;
  };
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) { const f32x4 a = *reinterpret_cast<const f32x4*>(px + s);
                                  const f32x4 g = *reinterpret_cast<const f32x4*>(pg + s);
                                  f32x4 o; o.x = one(a.x, g.x); o.y = one(a.y, g.y);
                                  o.z = one(a.z, g.z); o.w = one(a.w, g.w);
                                  *reinterpret_cast<f32x4*>(po + s) = o; },
                 [&](int64_t s) { po[s] = one(px[s], pg[s]); });
  if (dsum && c < ndsum) {       // bias gradient of the convolution that produced x: sum of dx
    const float w = crn_wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tsum = 0.f;
      for (int i = 0; i < kThreads / 64; ++i) tsum += red[i];
      atomicAdd(dsum + c, tsum);
    }
  }
}

// Channel-owner backward: reduce, then apply, in one workgroup per channel (second pass re-reads from L2).
template <bool VEC>
__global__ __launch_bounds__(kThreads) void bn_owner_bwd_kernel(
    const float* x, int64_t sBx, const float* dy, int64_t sBdy, int B, int64_t S, int C, int pre_relu,
    int post_relu, const float* gamma, const float* scale, const float* shift, const float* saved, float* dx,
    int64_t sBdx, float* dgamma, float* dbeta, int accumulate, float* dsum, int ndsum) {
  __shared__ double red[kThreads / 64];
  __shared__ float sm[2];
  __shared__ float redf[kThreads / 64];
  const int c = blockIdx.x;
  const float sc = scale[c], sh = shift[c], mu = saved[c], rstd = saved[C + c];
  Slice all; all.s0 = 0; all.s1 = S;
  double s1 = 0.0, s2 = 0.0;
  auto acc = [&](float xv, float gv) {
    if (pre_relu) xv = fmaxf(xv, 0.f);
    if (post_relu && !(xv * sc + sh > 0.f)) gv = 0.f;
    s1 += (double)gv;
    s2 += (double)gv * (double)((xv - mu) * rstd);
  };
  for (int b = 0; b < B; ++b) {
    const float* px = x + (int64_t)b * sBx + (int64_t)c * S;
    const float* pg = dy + (int64_t)b * sBdy + (int64_t)c * S;
    for_slice<VEC>(all,
                   [&](int64_t s) { const f32x4 a = *reinterpret_cast<const f32x4*>(px + s);
                                    const f32x4 g = *reinterpret_cast<const f32x4*>(pg + s);
                                    acc(a.x, g.x); acc(a.y, g.y); acc(a.z, g.z); acc(a.w, g.w); },
                   [&](int64_t s) { acc(px[s], pg[s]); });
  }
  const double t1 = crn_block_sum(s1, red);
  const double t2 = crn_block_sum(s2, red);
  if (threadIdx.x == 0) {
    const double count = (double)B * (double)S;
    sm[0] = (float)(t1 / count); sm[1] = (float)(t2 / count);
    const float r = saved[2 * C + c], d = saved[3 * C + c];
    const float dg = (float)(r * t2 + d * t1), db = (float)t1;
    if (accumulate) { dgamma[c] += dg; dbeta[c] += db; } else { dgamma[c] = dg; dbeta[c] = db; }
  }
  __syncthreads();
  const float mg = sm[0], mgx = sm[1];
  const float k = gamma[c] * saved[2 * C + c] * rstd;
  float lsum = 0.f;
  auto one = [&](float xr, float gv) -> float {
    const float xv = pre_relu ? fmaxf(xr, 0.f) : xr;
    if (post_relu && !(xv * sc + sh > 0.f)) gv = 0.f;
    float o = k * (gv - mg - (xv - mu) * rstd * mgx);
    if (pre_relu && !(xr > 0.f)) o = 0.f;
    lsum += o;
    return o;
  };
  for (int b = 0; b < B; ++b) {
    const float* px = x + (int64_t)b * sBx + (int64_t)c * S;
    const float* pg = dy + (int64_t)b * sBdy + (int64_t)c * S;
    float* po = dx + (int64_t)b * sBdx + (int64_t)c * S;
    for_slice<VEC>(all,
                   [&](int64_t s) { const f32x4 a = *reinterpret_cast<const f32x4*>(px + s);
                                    const f32x4 g = *reinterpret_cast<const f32x4*>(pg + s);
                                    f32x4 o; o.x = one(a.x, g.x); o.y = one(a.y, g.y);
                                    o.z = one(a.z, g.z); o.w = one(a.w, g.w);
                                    *reinterpret_cast<f32x4*>(po + s) = o; },
                   [&](int64_t s) { po[s] = one(px[s], pg[s]); });
  }
  if (dsum && c < ndsum) {       // bias gradient of the convolution that produced x: sum of dx
    const float w = crn_wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
      float tsum = 0.f;
      for (int i = 0; i < kThreads / 64; ++i) tsum += redf[i];
      dsum[c] = tsum;
    }
  }
}

// many channels, little data per channel: the channel-owner kernels (bench.py: 17.15 -> 16.95 ms per step)
// one workgroup per channel pays when there are enough channels to fill the chip or the channel fits the register
// kernels; a 64-channel map of 65536 values per channel (the stem) keeps 64 CUs busy with 1 MB each (75 us for the
// backward pass against 2 x ~15 us for the partial + apply pair on all CUs)
inline bool owner_form(int64_t S, int C, int B) {
  return C >= 64 && (int64_t)B * S <= 65536 && ((int64_t)B * S <= 16384 || C >= 256);
}

// ---- block tails ---------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(kThreads) void affine_add_relu_kernel(
    const float* x, const float* scale, const float* shift, const float* r, const float* rscale,
    const float* rshift, int64_t S, int64_t sBx, int64_t sBr, float* y_pre, int64_t sBpre,
    float* y, int64_t sBy, int relu) {
  crn_kernargs_now(x, scale, shift, r, rscale, rshift, S, sBx, sBr, y_pre, sBpre, y, sBy, relu);
  const int c = blockIdx.y, b = blockIdx.z;
  const float sc = scale ? scale[c] : 1.f, sh = shift ? shift[c] : 0.f;
  const float rsc = rscale ? rscale[c] : 1.f, rsh = rshift ? rshift[c] : 0.f;
  const float* px = x + (int64_t)b * sBx + (int64_t)c * S;
  const float* pr = r ? r + (int64_t)b * sBr + (int64_t)c * S : nullptr;
  float* pp = y_pre ? y_pre + (int64_t)b * sBpre + (int64_t)c * S : nullptr;
  float* py = y ? y + (int64_t)b * sBy + (int64_t)c * S : nullptr;
  auto one = [&](float xv, float rv) -> float { return xv * sc + sh + (pr ? rv * rsc + rsh : 0.f); };
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) {
                   const f32x4 a = *reinterpret_cast<const f32x4*>(px + s);
                   f32x4 q = (f32x4){0.f, 0.f, 0.f, 0.f};
                   if (pr) q = *reinterpret_cast<const f32x4*>(pr + s);
                   f32x4 o; o.x = one(a.x, q.x); o.y = one(a.y, q.y); o.z = one(a.z, q.z); o.w = one(a.w, q.w);
                   if (pp) *reinterpret_cast<f32x4*>(pp + s) = o;
                   if (py) {
                     if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                     *reinterpret_cast<f32x4*>(py + s) = o;
                   }
                 },
                 [&](int64_t s) {
                   const float o = one(px[s], pr ? pr[s] : 0.f);
                   if (pp) pp[s] = o;
                   if (py) py[s] = relu ? fmaxf(o, 0.f) : o;
                 });
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void relu_bwd_add_kernel(
    const float* dy, const float* y_pre, const float* dy2, int64_t S, int64_t sBdy, int64_t sBpre,
    int64_t sBdy2, float* dx, int64_t sBdx) {
  crn_kernargs_now(dy, y_pre, dy2, S, sBdy, sBpre, sBdy2, dx, sBdx);
  const int c = blockIdx.y, b = blockIdx.z;
  const float* pg = dy ? dy + (int64_t)b * sBdy + (int64_t)c * S : nullptr;
  const float* pp = y_pre + (int64_t)b * sBpre + (int64_t)c * S;
  const float* p2 = dy2 ? dy2 + (int64_t)b * sBdy2 + (int64_t)c * S : nullptr;
  float* po = dx + (int64_t)b * sBdx + (int64_t)c * S;
  auto one = [&](float g, float p, float g2) -> float { return (p > 0.f ? g : 0.f) + g2; };
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) {
                   f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f}, g2 = g;
                   if (pg) g = *reinterpret_cast<const f32x4*>(pg + s);
                   const f32x4 p = *reinterpret_cast<const f32x4*>(pp + s);
                   if (p2) g2 = *reinterpret_cast<const f32x4*>(p2 + s);
                   f32x4 o; o.x = one(g.x, p.x, g2.x); o.y = one(g.y, p.y, g2.y);
                   o.z = one(g.z, p.z, g2.z); o.w = one(g.w, p.w, g2.w);
                   *reinterpret_cast<f32x4*>(po + s) = o;
                 },
                 [&](int64_t s) { po[s] = one(pg ? pg[s] : 0.f, pp[s], p2 ? p2[s] : 0.f); });
}

template <bool VEC>
__global__ __launch_bounds__(kThreads) void bias_grad_partial_kernel(const float* dy, int64_t S,
                                                                      int64_t sB, double* ws) {
  __shared__ double red[kThreads / 64];
  const int c = blockIdx.y, b = blockIdx.z;
  const float* p = dy + (int64_t)b * sB + (int64_t)c * S;
  double s1 = 0.0;
  for_slice<VEC>(block_slice(S),
                 [&](int64_t s) { const f32x4 v = *reinterpret_cast<const f32x4*>(p + s);
                                  s1 += (double)v.x + (double)v.y + (double)v.z + (double)v.w; },
                 [&](int64_t s) { s1 += (double)p[s]; });
  const double t1 = crn_block_sum(s1, red);
  if (threadIdx.x == 0) {
    const int nparts = gridDim.x * gridDim.z;
    ws[(int64_t)c * nparts + (int64_t)b * gridDim.x + blockIdx.x] = t1;
  }
}

// small tensors (the skip-compress convs: <= 64 k elements per channel): one workgroup owns a channel and reduces it
// over batch and positions -- one launch instead of partial + final
__global__ __launch_bounds__(kThreads) void bias_grad_owner_kernel(const float* dy, int B, int64_t S, int64_t sB,
                                                                    float* db, int accumulate) {
  __shared__ double red[kThreads / 64];
  crn_kernargs_now(dy, B, S, sB, db, accumulate);
  const int c = blockIdx.x;
  double s1 = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* p = dy + (int64_t)b * sB + (int64_t)c * S;
    for (int64_t i = threadIdx.x; i < S; i += kThreads) s1 += (double)p[i];
  }
  const double t1 = crn_block_sum(s1, red);
  if (threadIdx.x == 0) db[c] = accumulate ? db[c] + (float)t1 : (float)t1;
}

__global__ void bias_grad_final_kernel(const double* ws, int nparts, int C, float* db, int accumulate) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // a wave per channel
  if (c >= C) return;
  double s = 0.0;
  for (int i = lane; i < nparts; i += 64) s += ws[(int64_t)c * nparts + i];
  s = crn_wave_sum(s);
  if (lane) return;
  if (accumulate) db[c] += (float)s; else db[c] = (float)s;
}

inline bool vec_ok(int64_t S, std::initializer_list<int64_t> strides, std::initializer_list<const void*> ptrs) {
  if (S & 3) return false;
  for (int64_t s : strides) if (s & 3) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

inline int nsplit_for(int64_t S, int C, int B) {
  // enough blocks to fill 256 CUs x 8, >= 4096 elements per block
  static const int kWant = getenv("CRN_BN_WANT") ? atoi(getenv("CRN_BN_WANT")) : 2048;       // tuning aid
  int64_t want = std::max<int64_t>(1, kWant / std::max(1, C * B));
  int64_t maxs = std::max<int64_t>(1, S / 4096);
  return (int)std::min<int64_t>(std::min(want, maxs), 64);
}

constexpr int kMaxParts = 64 * 64;   // nsplit(<=64) * B(<=64)

}  // namespace

extern "C" size_t crn_batch_renorm_workspace_bytes(int C) {
  return (size_t)C * kMaxParts * 2 * sizeof(double);
}

// tail != nullptr: the block tail (crn_affine_add_relu's arguments) is wanted in the same launch; *tail_done says whether
// the launch taken could do it (the register form of the owner kernel), otherwise the caller runs it separately
static int bn_stats_impl(const float* x, int B, int C, int64_t S, int64_t sB, int pre_relu,
                         const float* gamma, const float* beta, float* running_mean,
                         float* running_var, const int64_t* nbt, float eps, float momentum,
                         int training, float* scale, float* shift, float* saved,
                         double* ws, size_t ws_bytes, crnStream stream, const BnTail* tail, bool* tail_done) {
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || B > 64) return CRN_EINVAL;
  int nparts = 1;
  const bool reg_form = training && owner_form(S, C, B) && vec_ok(S, {sB}, {x}) && (int64_t)B * S <= 16384;
  // a split-K convolution left the partial sums of x pending (crn_splitk_defer): the register kernel adds them up
  const float* part = nullptr;
  int psplits = 0;
  {
    CrnSplitPending& pend = crn_splitk_pending();
    if (pend.active) {
      const crnView& py = pend.y;
      if (reg_form && pend.stream == st && py.base == x && py.B == B && py.C == C && (int64_t)py.D * py.H * py.W == S && sB == (int64_t)C * S) {
        part = pend.scratch; psplits = pend.splits; pend.active = false;
      } else {
        const int rcf = crn_splitk_flush(st);
        if (rcf != CRN_OK) return rcf;
      }
    }
  }
  if (reg_form) {
    const int per = (int)crn_cdiv((int64_t)B * S / 4, kThreads);       // float4 per thread
#define CRN_BN_STATS_REG(NV)                                                                                      \
  hipLaunchKernelGGL(bn_owner_stats_reg_kernel<NV>, dim3(C), dim3(kThreads), 0, st, const_cast<float*>(x), B, C, (int)(S / 4), sB, pre_relu, \
                     gamma, beta, running_mean, running_var, nbt, eps, momentum, scale, shift, saved, part, psplits)
#define CRN_BN_STATS_TAIL(NV)                                                                                     \
  hipLaunchKernelGGL((bn_owner_stats_reg_kernel<NV, true>), dim3(C), dim3(kThreads), 0, st, const_cast<float*>(x), B, C, (int)(S / 4), sB, pre_relu, \
                     gamma, beta, running_mean, running_var, nbt, eps, momentum, scale, shift, saved, part, psplits, *tail)
    if (tail && vec_ok(S, {tail->r ? tail->sBr : 0, tail->y_pre ? tail->sBpre : 0, tail->y ? tail->sBy : 0},
                       {tail->r, tail->y_pre, tail->y})) {
      if (per <= 1) CRN_BN_STATS_TAIL(1); else if (per <= 2) CRN_BN_STATS_TAIL(2); else if (per <= 4) CRN_BN_STATS_TAIL(4);
      else if (per <= 8) CRN_BN_STATS_TAIL(8); else CRN_BN_STATS_TAIL(16);
      *tail_done = true;
    } else
    if (per <= 1) CRN_BN_STATS_REG(1); else if (per <= 2) CRN_BN_STATS_REG(2); else if (per <= 4) CRN_BN_STATS_REG(4);
    else if (per <= 8) CRN_BN_STATS_REG(8); else CRN_BN_STATS_REG(16);
#undef CRN_BN_STATS_REG
#undef CRN_BN_STATS_TAIL
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  if (training && owner_form(S, C, B)) {
    if (vec_ok(S, {sB}, {x}))
      hipLaunchKernelGGL(bn_owner_stats_kernel<true>, dim3(C), dim3(kThreads), 0, st, x, B, C, S, sB, pre_relu, gamma,
                         beta, running_mean, running_var, nbt, eps, momentum, scale, shift, saved);
    else
      hipLaunchKernelGGL(bn_owner_stats_kernel<false>, dim3(C), dim3(kThreads), 0, st, x, B, C, S, sB, pre_relu, gamma,
                         beta, running_mean, running_var, nbt, eps, momentum, scale, shift, saved);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  if (training) {
    const int ns = nsplit_for(S, C, B);
    nparts = ns * B;
    if (ws_bytes < (size_t)C * nparts * 2 * sizeof(double)) return CRN_ENOMEM;
    dim3 grid(ns, C, B);
    if (vec_ok(S, {sB}, {x}))
      hipLaunchKernelGGL(bn_partial_kernel<true>, grid, dim3(kThreads), 0, st, x, S, sB, pre_relu, ws);
    else
      hipLaunchKernelGGL(bn_partial_kernel<false>, grid, dim3(kThreads), 0, st, x, S, sB, pre_relu, ws);
    CRN_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(crn_cdiv(C, 4)), dim3(256), 0, st, ws, nparts, C,
                     (double)B * (double)S, gamma, beta, running_mean, running_var, nbt, eps, momentum,
                     training, scale, shift, saved);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_batch_renorm_stats(const float* x, int B, int C, int64_t S, int64_t sB, int pre_relu,
                                      const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, const int64_t* nbt, float eps, float momentum,
                                      int training, float* scale, float* shift, float* saved,
                                      double* ws, size_t ws_bytes, crnStream stream) {
  return bn_stats_impl(x, B, C, S, sB, pre_relu, gamma, beta, running_mean, running_var, nbt, eps, momentum, training,
                       scale, shift, saved, ws, ws_bytes, stream, nullptr, nullptr);
}

// The second half of crn_batch_renorm_stats alone: the partial sums ws[(c * nparts + i) * 2 + {0, 1}] = sum(x), sum(x^2)
// of part i came out of the launch that produced x (crn_stem_conv_fwd); count = elements per channel.
extern "C" int crn_batch_renorm_finalize(const double* ws, int nparts, int C, double count,
                                         const float* gamma, const float* beta, float* running_mean,
                                         float* running_var, const int64_t* nbt, float eps, float momentum,
                                         float* scale, float* shift, float* saved, crnStream stream) {
  CRN_ENTRY(stream);
  if (!ws || nparts < 1 || nparts > kMaxParts || C < 1 || !(count >= 1.0)) return CRN_EINVAL;
  static const int kWide = getenv("CRN_BN_FINALIZE_WIDE") ? atoi(getenv("CRN_BN_FINALIZE_WIDE")) : 512;      // parts per channel from which on
  if (kWide > 0 && nparts >= kWide && (((uintptr_t)ws) & 15) == 0)
    hipLaunchKernelGGL(bn_finalize_wide_kernel, dim3(C), dim3(kThreads), 0, (hipStream_t)stream, ws, nparts, C, count,
                       gamma, beta, running_mean, running_var, nbt, eps, momentum, scale, shift, saved);
  else
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(crn_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, ws, nparts, C, count,
                     gamma, beta, running_mean, running_var, nbt, eps, momentum, 1, scale, shift, saved);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

int crn_affine_add_relu_impl(const float* x, const float* scale, const float* shift, const float* r, const float* rscale,
                             const float* rshift, int B, int C, int64_t S, int64_t sB_x, int64_t sB_r, float* y_pre,
                             int64_t sB_pre, float* y, int64_t sB_y, int relu, crnStream stream);

extern "C" int crn_batch_renorm_stats_tail(const float* x, int B, int C, int64_t S, int64_t sB,
                                           const float* gamma, const float* beta, float* running_mean,
                                           float* running_var, const int64_t* nbt, float eps, float momentum,
                                           int training, float* scale, float* shift, float* saved,
                                           double* ws, size_t ws_bytes,
                                           const float* r, const float* rscale, const float* rshift, int64_t sB_r,
                                           float* y_pre, int64_t sB_pre, float* y, int64_t sB_y, int relu,
                                           float* y2, int W, crnStream stream) {
  if (!y && !y_pre) return CRN_EINVAL;      // (no CRN_ENTRY: a pending split-K sum of x is taken over below)
  if (y2 && (!y || W < 1 || S % W || sB_y != (int64_t)C * S)) return CRN_EINVAL;
  const int H = y2 ? (int)(S / W) : 0, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const bool y2_fused = y2 && W % 4 == 0 && (((uintptr_t)y2) & 7) == 0;      // (a float4 inside one row, 8-byte stores)
  const BnTail tail{r, rscale, rshift, sB_r, y_pre, sB_pre, y, sB_y, relu, y2_fused ? y2 : nullptr, (int64_t)C * H2 * W2, W, W2, H2};
  bool done = false;
  const int rc = bn_stats_impl(x, B, C, S, sB, 0, gamma, beta, running_mean, running_var, nbt, eps, momentum, training,
                               scale, shift, saved, ws, ws_bytes, stream, &tail, &done);
  if (rc != CRN_OK) return rc;
  if (!done) {
    const int rc2 = crn_affine_add_relu_impl(x, scale, shift, r, rscale, rshift, B, C, S, sB, sB_r, y_pre, sB_pre, y, sB_y, relu, stream);
    if (rc2 != CRN_OK) return rc2;
  }
  if (y2 && !(done && y2_fused)) return crn_stride2_gather(y, y2, B, C, H2, W2, H, W, stream);
  return CRN_OK;
}

extern "C" int crn_batch_renorm_eval_affine(const float* params, const float* buffers, const int32_t* table,
                                            int n, float eps, float* scale, float* shift, crnStream stream) {
  CRN_ENTRY(stream);
  if (n < 0 || (n > 0 && (!params || !buffers || !table || !scale || !shift))) return CRN_EINVAL;
  if (n == 0) return CRN_OK;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(crn_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, params,
                     buffers, table, n, eps, scale, shift);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

int crn_relu_bwd_add_impl(const float* dy, const float* y_pre, const float* dy2, int B, int C, int64_t S, int64_t sB_dy,
                          int64_t sB_pre, int64_t sB_dy2, float* dx, int64_t sB_dx, crnStream stream);

// head != nullptr: dy is to be formed first from (g, act, g2) -- in the same launch where a workgroup owns a channel in
// registers, by crn_relu_bwd_add's kernel otherwise
static int bn_bwd_impl(const float* x, int64_t sB_x, const float* dy, int64_t sB_dy,
                       int B, int C, int64_t S, int pre_relu, int post_relu,
                       const float* gamma, const float* scale, const float* shift,
                       const float* saved, float* dx, int64_t sB_dx, float* dgamma,
                       float* dbeta, int accumulate, float* dsum, int ndsum, double* ws,
                       size_t ws_bytes, crnStream stream, const BnHead* head) {
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || B > 64) return CRN_EINVAL;
  const bool v = vec_ok(S, {sB_x, sB_dy, sB_dx}, {x, dy, dx});
  // The register-resident owner kernel keeps 16 float4 of x and of dy per thread when a channel has up to 16384 elements: 359
  // registers, i.e. a workgroup that only fits on a CU with NO wave of a weight-gradient kernel (206-250 registers x 2 waves per
  // SIMD) on it.  In the decoder's backward those run beside the data-gradient chain all the time: stage 4's two norms (64 / 112
  // channels = workgroups) took 125-150 us each inside the step against ~35 us alone.  The two-pass form (50 / 46 registers, many
  // workgroups) slots in beside them: step 7.30 -> 7.19 ms.  Applies to the pre-ReLU norms (the decoder's; the encoder's run
  // beside small weight gradients).  CRN_BN_BWD_DEC16: 0 the register form as before, 1 the streaming owner kernel, 2 (default)
  // two passes, 3 two passes for every norm of that size; CRN_BN_BWD_DEC_MIN: elements per channel from which it applies.
  static const int dec16 = getenv("CRN_BN_BWD_DEC16") ? atoi(getenv("CRN_BN_BWD_DEC16")) : 2;
  static const int dec_min = getenv("CRN_BN_BWD_DEC_MIN") ? atoi(getenv("CRN_BN_BWD_DEC_MIN")) : 8192;
  const bool big_dec = dec16 && (pre_relu || dec16 == 3) && !head && (int64_t)B * S > dec_min;
  const bool owner_ok = owner_form(S, C, B) && !(big_dec && dec16 >= 2);
  const bool reg_form = owner_ok && v && (int64_t)B * S <= 16384 && !big_dec;
  if (head) {
    const int rcf = crn_splitk_flush(st);          // (dy is formed here: nothing pending can be meant for this call)
    if (rcf != CRN_OK) return rcf;
    const bool hv = vec_ok(S, {head->sBg, head->sBact, head->g2 ? head->sBg2 : 0}, {head->g, head->act, head->g2});
    BnHead expanded = *head;
    if (head->gc && !(reg_form && hv && head->W % 4 == 0 && (((uintptr_t)head->gc) & 7) == 0)) {
      // the launch below reads the expanded gradient: expand it into the caller's buffer first
      const int H = (int)(S / head->W);
      const int rcs = crn_stride2_scatter(head->gc, const_cast<float*>(head->g), B, C, head->H2, head->W2, H, head->W, stream);
      if (rcs != CRN_OK) return rcs;
      expanded.gc = nullptr;
      head = &expanded;
    }
    if (reg_form && hv) {
      const int per = (int)crn_cdiv((int64_t)B * S / 4, kThreads);
#define CRN_BN_BWD_HEAD(NV)                                                                                         \
  hipLaunchKernelGGL((bn_owner_bwd_reg_kernel<NV, true>), dim3(C), dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, B, (int)(S / 4), C, \
                     pre_relu, post_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum, \
                     (const float*)nullptr, 0, *head)
      if (per <= 1) CRN_BN_BWD_HEAD(1); else if (per <= 2) CRN_BN_BWD_HEAD(2); else if (per <= 4) CRN_BN_BWD_HEAD(4);
      else if (per <= 8) CRN_BN_BWD_HEAD(8); else CRN_BN_BWD_HEAD(16);
#undef CRN_BN_BWD_HEAD
      CRN_CHECK_LAUNCH();
      return CRN_OK;
    }
    const int rch = crn_relu_bwd_add_impl(head->g, head->act, head->g2, B, C, S, head->sBg, head->sBact, head->sBg2,
                                          const_cast<float*>(dy), sB_dy, stream);
    if (rch != CRN_OK) return rch;
  }
  const float* part = nullptr;
  int psplits = 0;
  {
    CrnSplitPending& pend = crn_splitk_pending();
    if (pend.active) {
      const crnView& py = pend.y;
      if (reg_form && pend.stream == st && py.base == dy && py.B == B && py.C == C && (int64_t)py.D * py.H * py.W == S && sB_dy == (int64_t)C * S) {
        part = pend.scratch; psplits = pend.splits; pend.active = false;
      } else {
        const int rcf = crn_splitk_flush(st);
        if (rcf != CRN_OK) return rcf;
      }
    }
  }
  if (reg_form) {
    const int per = (int)crn_cdiv((int64_t)B * S / 4, kThreads);       // float4 per thread (of x and of dy)
#define CRN_BN_BWD_REG(NV)                                                                                          \
  hipLaunchKernelGGL(bn_owner_bwd_reg_kernel<NV>, dim3(C), dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, B, (int)(S / 4), C, \
                     pre_relu, post_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum, \
                     part, psplits)
    if (per <= 1) CRN_BN_BWD_REG(1); else if (per <= 2) CRN_BN_BWD_REG(2); else if (per <= 4) CRN_BN_BWD_REG(4);
    else if (per <= 8) CRN_BN_BWD_REG(8); else CRN_BN_BWD_REG(16);
#undef CRN_BN_BWD_REG
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  if (owner_ok) {
    if (v)
      hipLaunchKernelGGL(bn_owner_bwd_kernel<true>, dim3(C), dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, B, S, C, pre_relu,
                         post_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
    else
      hipLaunchKernelGGL(bn_owner_bwd_kernel<false>, dim3(C), dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, B, S, C, pre_relu,
                         post_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  const int ns = nsplit_for(S, C, B);
  const int nparts = ns * B;
  if (ws_bytes < (size_t)C * nparts * 2 * sizeof(double)) return CRN_ENOMEM;
  dim3 grid(ns, C, B);
  // deterministic mode: the ns * B workgroups of a channel add their share of sum(dx) atomically below; instead the
  // sum is taken afterwards by the ordered two-level reduction of crn_bias_grad over dx
  float* const dsum_det = (dsum && ndsum > 0 && crn_deterministic()) ? dsum : nullptr;
  if (dsum_det) dsum = nullptr;
  if (v)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, post_relu, scale, shift, saved, ws, dsum, ndsum);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, post_relu, scale, shift, saved, ws, dsum, ndsum);
  CRN_CHECK_LAUNCH();
  if (v)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, post_relu, gamma, scale, shift, saved, ws, nparts,
                       (double)B * (double)S, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, post_relu, gamma, scale, shift, saved, ws, nparts,
                       (double)B * (double)S, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
  CRN_CHECK_LAUNCH();
  if (dsum_det) {
    dim3 gridb(ns, ndsum, B);
    if (vec_ok(S, {sB_dx}, {dx}))
      hipLaunchKernelGGL(bias_grad_partial_kernel<true>, gridb, dim3(kThreads), 0, st, dx, S, sB_dx, ws);
    else
      hipLaunchKernelGGL(bias_grad_partial_kernel<false>, gridb, dim3(kThreads), 0, st, dx, S, sB_dx, ws);
    CRN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(crn_cdiv(ndsum, 4)), dim3(256), 0, st, ws, nparts, ndsum, dsum_det, 0);
    CRN_CHECK_LAUNCH();
  }
  return CRN_OK;
}

extern "C" int crn_batch_renorm_bwd(const float* x, int64_t sB_x, const float* dy, int64_t sB_dy,
                                    int B, int C, int64_t S, int pre_relu, int post_relu,
                                    const float* gamma, const float* scale, const float* shift,
                                    const float* saved, float* dx, int64_t sB_dx, float* dgamma,
                                    float* dbeta, int accumulate, float* dsum, int ndsum, double* ws,
                                    size_t ws_bytes, crnStream stream) {
  return bn_bwd_impl(x, sB_x, dy, sB_dy, B, C, S, pre_relu, post_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta,
                     accumulate, dsum, ndsum, ws, ws_bytes, stream, nullptr);
}

extern "C" int crn_batch_renorm_bwd_apply(const float* x, int64_t sB_x, const float* dy, int64_t sB_dy,
                                          int B, int C, int64_t S, int pre_relu,
                                          const float* gamma, const float* scale, const float* shift,
                                          const float* saved, float* dx, int64_t sB_dx, float* dgamma,
                                          float* dbeta, int accumulate, float* dsum, int ndsum, double* ws,
                                          size_t ws_bytes, int nparts, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || B > 64 || nparts < 1 || !ws) return CRN_EINVAL;
  if (ws_bytes < (size_t)C * nparts * 2 * sizeof(double)) return CRN_ENOMEM;
  { const int rcf = crn_splitk_flush(st); if (rcf != CRN_OK) return rcf; }
  const int ns = nsplit_for(S, C, B);
  dim3 grid(ns, C, B);
  // deterministic mode: sum(dx) by the ordered two-level reduction over dx afterwards (as in crn_batch_renorm_bwd); it re-uses
  // `ws` once the apply kernel has read the partial sums
  float* const dsum_det = (dsum && ndsum > 0 && crn_deterministic()) ? dsum : nullptr;
  if (dsum_det) dsum = nullptr;
  if (vec_ok(S, {sB_x, sB_dy, sB_dx}, {x, dy, dx}))
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, 0, gamma, scale, shift, saved, ws, nparts,
                       (double)B * (double)S, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(kThreads), 0, st, x, sB_x, dy, sB_dy, S, C,
                       pre_relu, 0, gamma, scale, shift, saved, ws, nparts,
                       (double)B * (double)S, dx, sB_dx, dgamma, dbeta, accumulate, dsum, ndsum);
  CRN_CHECK_LAUNCH();
  if (dsum_det) {
    const int nparts2 = ns * B;
    if (ws_bytes < (size_t)ndsum * nparts2 * sizeof(double)) return CRN_ENOMEM;
    dim3 gridb(ns, ndsum, B);
    if (vec_ok(S, {sB_dx}, {dx}))
      hipLaunchKernelGGL(bias_grad_partial_kernel<true>, gridb, dim3(kThreads), 0, st, dx, S, sB_dx, ws);
    else
      hipLaunchKernelGGL(bias_grad_partial_kernel<false>, gridb, dim3(kThreads), 0, st, dx, S, sB_dx, ws);
    CRN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(crn_cdiv(ndsum, 4)), dim3(256), 0, st, ws, nparts2, ndsum, dsum_det, 0);
    CRN_CHECK_LAUNCH();
  }
  return CRN_OK;
}

extern "C" int crn_batch_renorm_bwd_head(const float* x, int64_t sB_x, float* dy, int64_t sB_dy,
                                         const float* g, int64_t sB_g, const float* act, int64_t sB_act,
                                         const float* g2, int64_t sB_g2,
                                         int B, int C, int64_t S, const float* gamma, const float* scale,
                                         const float* shift, const float* saved, float* dx, int64_t sB_dx,
                                         float* dgamma, float* dbeta, int accumulate, float* dsum, int ndsum,
                                         double* ws, size_t ws_bytes, const float* g_compact, int W, crnStream stream) {
  if (!g || !act || !dy) return CRN_EINVAL;
  if (g_compact && (W < 1 || S % W || sB_g != (int64_t)C * S)) return CRN_EINVAL;
  const int H = g_compact ? (int)(S / W) : 0, H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const BnHead head{g, sB_g, act, sB_act, g2, sB_g2, g_compact, (int64_t)C * H2 * W2, W, W2, H2};
  return bn_bwd_impl(x, sB_x, dy, sB_dy, B, C, S, 0, 0, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta,
                     accumulate, dsum, ndsum, ws, ws_bytes, stream, &head);
}

extern "C" int crn_affine_add_relu(const float* x, const float* scale, const float* shift,
                                   const float* r, const float* rscale, const float* rshift,
                                   int B, int C, int64_t S, int64_t sB_x, int64_t sB_r,
                                   float* y_pre, int64_t sB_pre, float* y, int64_t sB_y, int relu,
                                   crnStream stream) {
  CRN_ENTRY(stream);
  return crn_affine_add_relu_impl(x, scale, shift, r, rscale, rshift, B, C, S, sB_x, sB_r, y_pre, sB_pre, y, sB_y, relu, stream);
}
int crn_affine_add_relu_impl(const float* x, const float* scale, const float* shift, const float* r, const float* rscale,
                             const float* rshift, int B, int C, int64_t S, int64_t sB_x, int64_t sB_r, float* y_pre,
                             int64_t sB_pre, float* y, int64_t sB_y, int relu, crnStream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || (!y && !y_pre)) return CRN_EINVAL;
  dim3 grid(nsplit_for(S, C, B), C, B);
  if (vec_ok(S, {sB_x, r ? sB_r : 0, y_pre ? sB_pre : 0, y ? sB_y : 0}, {x, r, y_pre, y}))
    hipLaunchKernelGGL(affine_add_relu_kernel<true>, grid, dim3(kThreads), 0, st, x, scale, shift, r, rscale,
                       rshift, S, sB_x, sB_r, y_pre, sB_pre, y, sB_y, relu);
  else
    hipLaunchKernelGGL(affine_add_relu_kernel<false>, grid, dim3(kThreads), 0, st, x, scale, shift, r, rscale,
                       rshift, S, sB_x, sB_r, y_pre, sB_pre, y, sB_y, relu);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_relu_bwd_add(const float* dy, const float* y_pre, const float* dy2, int B, int C,
                                int64_t S, int64_t sB_dy, int64_t sB_pre, int64_t sB_dy2, float* dx,
                                int64_t sB_dx, crnStream stream) {
  CRN_ENTRY(stream);
  return crn_relu_bwd_add_impl(dy, y_pre, dy2, B, C, S, sB_dy, sB_pre, sB_dy2, dx, sB_dx, stream);
}
int crn_relu_bwd_add_impl(const float* dy, const float* y_pre, const float* dy2, int B, int C, int64_t S, int64_t sB_dy,
                          int64_t sB_pre, int64_t sB_dy2, float* dx, int64_t sB_dx, crnStream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || !y_pre || !dx) return CRN_EINVAL;
  dim3 grid(nsplit_for(S, C, B), C, B);
  if (vec_ok(S, {dy ? sB_dy : 0, sB_pre, dy2 ? sB_dy2 : 0, sB_dx}, {dy, y_pre, dy2, dx}))
    hipLaunchKernelGGL(relu_bwd_add_kernel<true>, grid, dim3(kThreads), 0, st, dy, y_pre, dy2, S, sB_dy,
                       sB_pre, sB_dy2, dx, sB_dx);
  else
    hipLaunchKernelGGL(relu_bwd_add_kernel<false>, grid, dim3(kThreads), 0, st, dy, y_pre, dy2, S, sB_dy,
                       sB_pre, sB_dy2, dx, sB_dx);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_bias_grad(const float* dy, int B, int C, int64_t S, int64_t sB, float* db,
                             int accumulate, double* ws, size_t ws_bytes, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || S < 1 || B > 64) return CRN_EINVAL;
  if ((int64_t)B * S <= 65536) {
    hipLaunchKernelGGL(bias_grad_owner_kernel, dim3(C), dim3(kThreads), 0, st, dy, B, S, sB, db, accumulate);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  const int ns = nsplit_for(S, C, B);
  const int nparts = ns * B;
  if (ws_bytes < (size_t)C * nparts * sizeof(double)) return CRN_ENOMEM;
  dim3 grid(ns, C, B);
  if (vec_ok(S, {sB}, {dy}))
    hipLaunchKernelGGL(bias_grad_partial_kernel<true>, grid, dim3(kThreads), 0, st, dy, S, sB, ws);
  else
    hipLaunchKernelGGL(bias_grad_partial_kernel<false>, grid, dim3(kThreads), 0, st, dy, S, sB, ws);
  CRN_CHECK_LAUNCH();
  hipLaunchKernelGGL(bias_grad_final_kernel, dim3(crn_cdiv(C, 4)), dim3(256), 0, st, ws, nparts, C, db,
                     accumulate);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
