// Fused softmax + IoU / cross-entropy losses (forward AND gradient) and the
// eval epilogue (argmax + confusion matrix).
// Reference: model/losses.py:19-160 (iou_agnostic, iou_fgbg, xent,
// xent_times_iou_*), evaluation_results.py:40-51, voxel_metrics.py:33-58.
// The reference materialises one-hot and several [B,C,128^3] temporaries
// (3.0 s CPU at B=4,C=14); here: two streaming passes over the logits, fp64
// block reductions, no temporaries.
#include "crn_common.h"
#include <algorithm>
#include <cmath>

namespace {

constexpr int kThreads = 256;
constexpr int kNQ = 6;   // per-sample sums: I_fg, U_fg, I_ag, U_ag, X, and the number of labels outside [0, C) (the status word: it
                         // travels with the partial sums, so no memset -- a launch of its own on the step's chain -- precedes pass 1)

// per-voxel softmax over CT (>= C) classes held in registers
template <int CT>
struct Soft {
  float s[CT];
  float logz;   // log sum exp (relative to max) + max
  __device__ __forceinline__ void compute(const float* l, int64_t S, int C) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CT; ++c) { s[c] = c < C ? l[c * S] : -INFINITY; m = fmaxf(m, s[c]); }
    float z = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) { s[c] = c < C ? expf(s[c] - m) : 0.f; z += s[c]; }
    const float inv = 1.0f / z;
#pragma unroll
    for (int c = 0; c < CT; ++c) s[c] *= inv;
    logz = m + logf(z);
  }
};

template <int CT>
__global__ __launch_bounds__(kThreads) void loss_pass1_kernel(const float* logits, const int32_t* gt,
                                                              const float* weights, int C, int64_t S, double* part) {
  __shared__ double red[kThreads / 64];
  const int b = blockIdx.y;
  const float* lb = logits + (int64_t)b * C * S;
  const int32_t* gb = gt + (int64_t)b * S;
  const float* wb = weights ? weights + (int64_t)b * S : nullptr;
  double q[kNQ] = {0, 0, 0, 0, 0, 0};
  for (int64_t v = blockIdx.x * (int64_t)kThreads + threadIdx.x; v < S; v += (int64_t)gridDim.x * kThreads) {
    Soft<CT> sm;
    sm.compute(lb + v, S, C);
    int g = gb[v];
    if (g < 0 || g >= C) { q[5] += 1.0; g = 0; }       // F.one_hot / cross_entropy raise (losses.py:36,131); no OOB read here
    const float w = wb ? wb[v] : 1.f;                   // per-voxel loss weights (losses.py:47-49,99-102,134-136)
    float pfg = 0.f;
#pragma unroll
    for (int c = 1; c < CT; ++c) pfg += sm.s[c];
    const float gf = g >= 1 ? 1.f : 0.f;
    q[0] += (double)(fminf(gf, pfg) * w);
    q[1] += (double)(fmaxf(gf, pfg) * w);
    float ia = 0.f, ua = 0.f;
#pragma unroll
    for (int c = 1; c < CT; ++c) {
      if (c < C) {
        if (c == g) { ia += sm.s[c] * ((float)(C - 1) * w); ua += (float)(C - 1) * w; }   // min(1,p)=p, max(1,p)=1
        else ua += sm.s[c] * w;
      }
    }
    q[2] += (double)ia; q[3] += (double)ua;
    q[4] += (double)((sm.logz - lb[v + (int64_t)g * S]) * w);
  }
  for (int k = 0; k < kNQ; ++k) {
    const double t = crn_block_sum(q[k], red);
    if (threadIdx.x == 0) part[((int64_t)b * gridDim.x + blockIdx.x) * kNQ + k] = t;
  }
}

// coef layout (floats): [0]=loss, [1]=kIouFg, [2]=kIouAg, [3]=kX, then per b: I_fg,U_fg,I_ag,U_ag
// One workgroup of 256 threads: wave w reduces the (sample, quantity) pairs w, w+4, ... (the per-block partials of
// a pair are summed in a fixed order: lane-strided, then the wave tree), thread 0 combines them.  (A single wave
// walking all B*5 pairs one after the other took 25 us of the step's critical path.)
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const double* part, int nblk, int B, int64_t S, int kind,
                                                            float grad_scale, float* loss, float* coef, int* bad_label) {
  __shared__ double sums[64 * kNQ];                      // B <= 64
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pair = wave; pair < B * kNQ; pair += (int)(blockDim.x >> 6)) {     // 16 waves: 20 sums in two rounds
    const int b = pair / kNQ, k = pair - b * kNQ;
    double s = 0.0;
    for (int i = lane; i < nblk; i += 64) s += part[((int64_t)b * nblk + i) * kNQ + k];
    s = crn_wave_sum(s);
    if (lane == 0) sums[pair] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double iou_fg = 0.0, iou_ag = 0.0, xs = 0.0, nbad = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* q = sums + b * kNQ;
    nbad += q[5];
    // losses.py:57,110: union==0 -> divide by 1
    const float ifg = (float)q[0], ufg = (float)q[1] == 0.f ? 1.f : (float)q[1];
    const float iag = (float)q[2], uag = (float)q[3] == 0.f ? 1.f : (float)q[3];
    iou_fg += (double)(ifg / ufg); iou_ag += (double)(iag / uag);
    xs += q[4];
    coef[4 + b * 4 + 0] = ifg; coef[4 + b * 4 + 1] = ufg; coef[4 + b * 4 + 2] = iag; coef[4 + b * 4 + 3] = uag;
  }
  const float Lfg = 1.f - (float)(iou_fg / B), Lag = 1.f - (float)(iou_ag / B);
  const float X = (float)(xs / ((double)B * (double)S));
  float L = 0.f, kfg = 0.f, kag = 0.f, kx = 0.f;
  switch (kind) {
    case 0: L = Lfg; kfg = 1.f; break;
    case 1: L = (1.f + Lag) * (1.f + X); kag = 1.f + X; kx = 1.f + Lag; break;
    case 2: L = Lag; kag = 1.f; break;
    case 3: L = X; kx = 1.f; break;
    case 4: L = (1.f + Lfg) * (1.f + X); kfg = 1.f + X; kx = 1.f + Lfg; break;
  }
  *bad_label = nbad > 0.0 ? 1 : 0;
  loss[0] = L;
  coef[0] = L; coef[1] = kfg * grad_scale / B; coef[2] = kag * grad_scale / B;
  coef[3] = kx * grad_scale / (float)((double)B * (double)S);
}

template <int CT>
__global__ __launch_bounds__(kThreads) void loss_pass2_kernel(const float* logits, const int32_t* gt,
                                                              const float* weights, int C, int64_t S,
                                                              const float* coef, float* dlogits) {
  const int b = blockIdx.y;
  const float* lb = logits + (int64_t)b * C * S;
  const int32_t* gb = gt + (int64_t)b * S;
  const float* wb = weights ? weights + (int64_t)b * S : nullptr;
  float* db = dlogits + (int64_t)b * C * S;
  const float kfg0 = coef[1], kag0 = coef[2], kx0 = coef[3];
  const float ifg = coef[4 + b * 4 + 0], ufg = coef[4 + b * 4 + 1];
  const float iag = coef[4 + b * 4 + 2], uag = coef[4 + b * 4 + 3];
  for (int64_t v = blockIdx.x * (int64_t)kThreads + threadIdx.x; v < S; v += (int64_t)gridDim.x * kThreads) {
    Soft<CT> sm;
    sm.compute(lb + v, S, C);
    int g = gb[v];
    if (g < 0 || g >= C) g = 0;
    const float w = wb ? wb[v] : 1.f;                   // every per-voxel term of the three sums carries w
    const float kfg = kfg0 * w, kag = kag0 * w, kx = kx0 * w;
    float d[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) d[c] = 0.f;
    if (kfg0 != 0.f) {
      float pfg = 0.f;
#pragma unroll
      for (int c = 1; c < CT; ++c) pfg += sm.s[c];
      // dL/dp_fg = -(1/B) ( g/U - (1-g) I/U^2 )
      const float dp = g >= 1 ? -kfg / ufg : kfg * ifg / (ufg * ufg);
#pragma unroll
      for (int c = 0; c < CT; ++c) d[c] += dp * sm.s[c] * ((c >= 1 ? 1.f : 0.f) - pfg);
    }
    if (kag0 != 0.f) {
      float qv[CT];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        qv[c] = 0.f;
        if (c >= 1 && c < C) qv[c] = (c == g) ? -kag * (float)(C - 1) / uag : kag * iag / (uag * uag);
        dot += qv[c] * sm.s[c];
      }
#pragma unroll
      for (int c = 0; c < CT; ++c) d[c] += sm.s[c] * (qv[c] - dot);
    }
    if (kx0 != 0.f) {
#pragma unroll
      for (int c = 0; c < CT; ++c) d[c] += kx * (sm.s[c] - (c == g ? 1.f : 0.f));
    }
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < C) db[v + (int64_t)c * S] = d[c];
  }
}

template <int CT>
__global__ __launch_bounds__(kThreads) void argmax_confusion_kernel(const float* logits, const int32_t* gt,
                                                                    int C, int64_t S, int32_t* labels,
                                                                    unsigned long long* cm) {
  extern __shared__ int hist[];
  for (int i = threadIdx.x; i < C * C; i += kThreads) hist[i] = 0;
  __syncthreads();
  const int b = blockIdx.y;
  const float* lb = logits + (int64_t)b * C * S;
  for (int64_t v = blockIdx.x * (int64_t)kThreads + threadIdx.x; v < S; v += (int64_t)gridDim.x * kThreads) {
    float best = lb[v];
    int bi = 0;
#pragma unroll
    for (int c = 1; c < CT; ++c)
      if (c < C) { const float x = lb[v + (int64_t)c * S]; if (x > best) { best = x; bi = c; } }
    if (labels) labels[(int64_t)b * S + v] = bi;
    if (gt) {
      const int g = gt[(int64_t)b * S + v];
      if (g >= 0 && g < C) atomicAdd(&hist[g * C + bi], 1);
    }
  }
  __syncthreads();
  if (gt)
    for (int i = threadIdx.x; i < C * C; i += kThreads)
      if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

inline int loss_nblk(int64_t S) { return (int)std::min<int64_t>(std::max<int64_t>(1, (S + kThreads * 8 - 1) / (kThreads * 8)), 512); }

}  // namespace

extern "C" size_t crn_loss_workspace_bytes(int B, int C) {
  (void)C;
  return (size_t)B * 512 * kNQ * sizeof(double) + (size_t)(4 + 4 * B) * sizeof(float) + 64;   // last 64 B: status word
}

extern "C" int crn_loss_fwd_bwd(int kind, const float* logits, const int32_t* gt, const float* weights, int B, int C,
                                int64_t S, float* loss, float* dlogits, float grad_scale, void* workspace,
                                size_t workspace_bytes, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (kind < 0 || kind > 4 || B < 1 || C < 2 || C > 32 || S < 1 || !logits || !gt || !loss) return CRN_EINVAL;
  if (workspace_bytes < crn_loss_workspace_bytes(B, C)) return CRN_ENOMEM;
  const int nblk = loss_nblk(S);
  double* part = reinterpret_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(part + (size_t)B * 512 * kNQ);
  int* bad_label = reinterpret_cast<int*>(coef + 4 + 4 * B + (16 - (4 + 4 * B) % 16) % 16);   // crn_loss_status reads it
  dim3 grid(nblk, B);
#define CRN_LOSS_P1(CT) hipLaunchKernelGGL(loss_pass1_kernel<CT>, grid, dim3(kThreads), 0, st, logits, gt, weights, C, S, part)
  if (C <= 2) CRN_LOSS_P1(2); else if (C <= 4) CRN_LOSS_P1(4); else if (C <= 8) CRN_LOSS_P1(8);
  else if (C <= 16) CRN_LOSS_P1(16); else CRN_LOSS_P1(32);
#undef CRN_LOSS_P1
  CRN_CHECK_LAUNCH();
  if (B > 64) return CRN_EINVAL;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nblk, B, S, kind, grad_scale, loss, coef, bad_label);
  CRN_CHECK_LAUNCH();
  if (dlogits) {
#define CRN_LOSS_P2(CT) hipLaunchKernelGGL(loss_pass2_kernel<CT>, grid, dim3(kThreads), 0, st, logits, gt, weights, C, S, coef, dlogits)
    if (C <= 2) CRN_LOSS_P2(2); else if (C <= 4) CRN_LOSS_P2(4); else if (C <= 8) CRN_LOSS_P2(8);
    else if (C <= 16) CRN_LOSS_P2(16); else CRN_LOSS_P2(32);
#undef CRN_LOSS_P2
    CRN_CHECK_LAUNCH();
  }
  return CRN_OK;
}

// Device address of the status word of the last crn_loss_fwd_bwd on this workspace: != 0 when a label was
// outside [0, C) (such a label is computed as class 0; the reference raises inside F.one_hot).
extern "C" const int* crn_loss_status_ptr(void* workspace, int B) {
  double* part = reinterpret_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(part + (size_t)B * 512 * kNQ);
  return reinterpret_cast<int*>(coef + 4 + 4 * B + (16 - (4 + 4 * B) % 16) % 16);
}

extern "C" int crn_argmax_confusion(const float* logits, const int32_t* gt, int B, int C, int64_t S,
                                    int32_t* labels, int64_t* cm, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (B < 1 || C < 1 || C > 32 || S < 1 || !logits || (gt && !cm)) return CRN_EINVAL;
  dim3 grid(loss_nblk(S), B);
  const size_t sh = (size_t)C * C * sizeof(int);
#define CRN_AM(CT) hipLaunchKernelGGL(argmax_confusion_kernel<CT>, grid, dim3(kThreads), sh, st, logits, gt, C, S, labels, reinterpret_cast<unsigned long long*>(cm))
  if (C <= 2) CRN_AM(2); else if (C <= 4) CRN_AM(4); else if (C <= 8) CRN_AM(8);
  else if (C <= 16) CRN_AM(16); else CRN_AM(32);
#undef CRN_AM
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

// ---- multi-offset inference epilogue (super_resolution.py:92-125) -----------------------------
// pmf = softmax over classes of the logits of offset n = (iz*m + iy)*m + ix, written to the
// interleaved grid out[b][c][z*m+iz][y*m+iy][x*m+ix]: the reference's softmax + stack + reshape +
// 8-axis permute + reshape as one pass with coalesced writes.  m == 1 is a plain channel softmax.
namespace {
template <int CT>
__global__ __launch_bounds__(256) void softmax_superres_kernel(const float* __restrict__ logits, int m, int B, int C,
                                                               int D, int H, int W, float* __restrict__ out) {
  const int64_t So = (int64_t)D * m * H * m * W * m, S = (int64_t)D * H * W;
  const int b = blockIdx.y;
  for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < So; o += (int64_t)gridDim.x * blockDim.x) {
    const int Wo = W * m, Ho = H * m;
    const int X = (int)(o % Wo), Y = (int)((o / Wo) % Ho), Z = (int)(o / ((int64_t)Wo * Ho));
    const int n = ((Z % m) * m + (Y % m)) * m + (X % m);
    const int64_t s = ((int64_t)(Z / m) * H + (Y / m)) * W + (X / m);
    const float* src = logits + (((int64_t)n * B + b) * C) * S + s;
    float v[CT];
    float mx = -3.4e38f;
#pragma unroll
    for (int c = 0; c < CT; ++c) { v[c] = c < C ? src[(int64_t)c * S] : -3.4e38f; mx = fmaxf(mx, v[c]); }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CT; ++c) { v[c] = c < C ? expf(v[c] - mx) : 0.f; sum += v[c]; }
    float* dst = out + ((int64_t)b * C) * So + o;
#pragma unroll
    for (int c = 0; c < CT; ++c) if (c < C) dst[(int64_t)c * So] = v[c] / sum;
  }
}
}  // namespace

extern "C" int crn_softmax_superres(const float* logits, int m, int B, int C, int D, int H, int W, float* out,
                                    crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!logits || !out || m < 1 || B < 1 || C < 1 || C > 32 || D < 1 || H < 1 || W < 1) return CRN_EINVAL;
  const int64_t So = (int64_t)D * H * W * m * m * m;
  dim3 grid((unsigned)std::min<int64_t>(crn_cdiv(So, 256), 65535), (unsigned)B);
#define CRN_SM(CT) hipLaunchKernelGGL(softmax_superres_kernel<CT>, grid, dim3(256), 0, st, logits, m, B, C, D, H, W, out)
  if (C <= 2) CRN_SM(2); else if (C <= 4) CRN_SM(4); else if (C <= 8) CRN_SM(8);
  else if (C <= 16) CRN_SM(16); else CRN_SM(32);
#undef CRN_SM
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
