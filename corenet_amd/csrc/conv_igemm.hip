// fp32 MFMA implicit-GEMM convolution engine for gfx950 (CDNA4).
//
// One stride-1 window correlation over a logical NCDHW view; Conv2d / Conv3d /
// ConvTranspose3d forward + data-gradient + weight-gradient of the reference
// (resnet50.py:62-69,95-107,124; reconstruction_decoder.py:49-95;
// ray_traced_skip_connection.py:38) all map onto it through views and packed
// weights built by corenet_amd/model/conv_geometry.py.
//
// Design (MI355X-first, not a cuDNN/im2col translation):
//  * A workgroup (4 waves) owns a TDxTHxTW tile of output positions and an
//    N-tile of NSUB*16 output channels.  For each chunk of CC input channels it
//    stages the raw input PATCH (tile + window halo) and the weight chunk in
//    LDS once, then walks the window taps: every tap is an LDS *offset*, not a
//    new gather, so each HBM/L2 byte of the patch feeds kd*kh*kw*N MACs.
//  * v_mfma_f32_16x16x4_f32: rows = 16 output positions (an mh x mw sub-tile),
//    cols = 16 output channels, k = 4 input channels at one tap.  Exact fp32
//    (the reference is fp32 end to end), 157 TF/s peak.
//  * Register-staged software pipeline: the global loads of chunk c+1 are
//    issued into VGPRs before the MFMA loop of chunk c and written to LDS after
//    it, so HBM/L2 latency hides under the matrix pipe (cdna guide T14).
//  * LDS strides are padded to 16 (mod 32) banks so the two 32-lane halves of a
//    ds_read_b32 (k = 0,1 / 2,3) never collide.
//  * BatchRenorm-apply + ReLU of the producer are fused into the patch staging
//    (crnInTransform), so normalised activations are never written to HBM.
#include "crn_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int PREG = 32;    // staged patch floats per thread   (CC*PS   <= 256*PREG)
constexpr int WREG = 8;     // staged weight float4 per thread  (CC*T*NB <= 256*WREG*4)
constexpr int DREG = 32;    // staged dy floats per thread      (NB*npos <= 256*DREG)
constexpr int kMaxStage = 256 * PREG;

struct ConvGeom {
  crnView x, y;
  crnInTransform tr;
  const float* w;
  const float* bias;
  int Npad, bias_sB;
  int kd, kh, kw, pd, ph, pw, T;
  int TD, TH, TW;          // tile (positions)
  int mw, mh;              // M-subtile shape, mw*mh == 16
  int nsh, nsw;            // sub-tiles per tile along H, W
  int PD, PH, PW, PS, PSP; // patch dims, size, padded channel stride
  int WSP;                 // LDS weight channel stride
  int CC;                  // channels per chunk (multiple of 4)
  int tilesD, tilesH, tilesW;
  int nchunks, chunks_per_split;
  int mode;                // 0 store, 1 accumulate (rmw), 2 atomic add
  int lg2, npass;          // patch staging: plane padded to 2^lg2 slots, passes of 256 slots
  unsigned magic_PW, magic_PD, magic_T;
};

__device__ __forceinline__ int64_t view_chan(const crnView& v, int c) {
  return v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
}

// ---- patch staging: global -> registers (issue) and registers -> LDS (commit) ----
// The patch of one chunk is CC*PD planes of PH*PW elements.  Staging slots are laid out as
// planes padded to PLP = 2^lg2 elements so that slot -> (plane q, in-plane r) is shift/mask:
//   lg2 >= 8: q is wave-uniform (scalar unit) and r takes PLP/256 per-thread values,
//   lg2 <  8: r is a per-thread constant.
// Everything that depends only on r (ph, pw, h/w bounds, h/w address part) is loop invariant
// and hoisted by the compiler; per slot ~6 VALU remain for issue and ~8 for commit.
struct PatchDesc {
  crnView x;
  crnInTransform tr;
  int pd, ph, pw, PD, PH, PW, plane, lg2, PSP;
  unsigned magic_PW, magic_PD;       // ceil(2^20 / d)
};

typedef __amdgpu_buffer_rsrc_t crn_rsrc;   // buffer resource (V#)

__device__ __forceinline__ int mdiv(int x, unsigned magic) { return (int)(((unsigned)x * magic) >> 20); }

__device__ __forceinline__ crn_rsrc make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFF0u, 0x00020000);
}

// per-chunk channel tables in LDS: [0,64) element offsets, [64,128) BRN scale, [128,192) BRN shift
constexpr int kChTab = 192;
__device__ __forceinline__ void stage_choff(const crnView& v, const crnInTransform& tr, unsigned* choff, int c0,
                                            int nch) {
  const int t = threadIdx.x;
  if (t < nch) {
    const int c = min(c0 + t, v.C - 1);
    choff[t] = v.chan_off ? (unsigned)v.chan_off[c] : (unsigned)c * (unsigned)v.sC;
    if (tr.scale) {
      reinterpret_cast<float*>(choff)[64 + t] = tr.scale[c];
      reinterpret_cast<float*>(choff)[128 + t] = tr.shift[c];
    }
  }
}

// In-plane slot variant: everything that depends only on (thread, tile) and not on the plane.
// U = slots of 256 per padded plane (1, 2 or 4: compile time) -> U variants per thread; U == 0:
// planes smaller than 256 slots -> one variant, the plane index is per thread.
struct HWVar { int r; unsigned off; int in; };   // r = in-plane index (or -1), off = gh*sH + gw*sW

template <int U>
__device__ __forceinline__ void patch_hw(const PatchDesc& g, int h0, int w0, HWVar (&hw)[U ? U : 1]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < (U ? U : 1); ++u) {
    const int r = U ? tid + u * 256 : (tid & ((1 << g.lg2) - 1));
    const bool valid = r < g.plane;
    const int ph = mdiv(r, g.magic_PW), pw = r - ph * g.PW;
    const int gh = h0 + ph - g.ph, gw = w0 + pw - g.pw;
    hw[u].r = valid ? r : -1;
    hw[u].in = valid && (unsigned)gh < (unsigned)g.x.H && (unsigned)gw < (unsigned)g.x.W;
    hw[u].off = (unsigned)gh * (unsigned)g.x.sH + (unsigned)gw * (unsigned)g.x.sW;
  }
}

__device__ __forceinline__ unsigned chan_off_s(const crnView& v, int c) {   // c wave-uniform -> scalar load
  return v.chan_off ? (unsigned)v.chan_off[c] : (unsigned)c * (unsigned)v.sC;
}

template <int J, int U>
__device__ __forceinline__ void patch_issue_one(const PatchDesc& g, const unsigned* choff,
                                                const HWVar (&hw)[U ? U : 1], const crn_rsrc& rs, int nplanes,
                                                int c0, int d0, float& v) {
  unsigned goff = 0xFFFFFFFFu;               // outside the descriptor range -> the load returns 0
  if constexpr (U >= 1) {
    int q = J / U;
    asm volatile("" : "+s"(q));              // recompute per chunk on the scalar unit
    const HWVar& h = hw[J % U];
    const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
    const int c = c0 + cl, gd = d0 + pd - g.pd;
    if (q < nplanes && c < g.x.C && (unsigned)gd < (unsigned)g.x.D) {       // wave-uniform
      const unsigned sbase = chan_off_s(g.x, c) + (unsigned)gd * (unsigned)g.x.sD;
      goff = h.in ? (sbase + h.off) * 4u : 0xFFFFFFFFu;
    }
  } else {
    int jq = J * (256 >> g.lg2);
    asm volatile("" : "+s"(jq));
    const HWVar& h = hw[0];
    const int q = jq + ((int)threadIdx.x >> g.lg2);
    const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
    const int gd = d0 + pd - g.pd;
    const bool in = h.in && q < nplanes && (c0 + cl < g.x.C) && (unsigned)gd < (unsigned)g.x.D;
    if (in) goff = (choff[cl] + (unsigned)gd * (unsigned)g.x.sD + h.off) * 4u;
  }
  v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)goff, 0, 0));
}

template <int J, int U>
__device__ __forceinline__ void patch_commit_one(const PatchDesc& g, const unsigned* choff,
                                                 const HWVar (&hw)[U ? U : 1], float* ldsA, int nplanes, int c0,
                                                 int d0, float v) {
  if constexpr (U >= 1) {
    int q = J / U;
    asm volatile("" : "+s"(q));
    const HWVar& h = hw[J % U];
    if (q < nplanes) {                                   // wave-uniform
      const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
      const int c = c0 + cl, gd = d0 + pd - g.pd;
      if (g.tr.scale && c < g.x.C && (unsigned)gd < (unsigned)g.x.D) {
        const float sc = g.tr.scale[c], sh = g.tr.shift[c];           // scalar loads
        float t = v;
        if (g.tr.pre_relu) t = fmaxf(t, 0.f);
        t = t * sc + sh;
        if (g.tr.post_relu) t = fmaxf(t, 0.f);
        v = h.in ? t : v;                                // zero padding stays zero
      }
      if (h.r >= 0) ldsA[cl * g.PSP + pd * g.plane + h.r] = v;
    }
  } else {
    int jq = J * (256 >> g.lg2);
    asm volatile("" : "+s"(jq));
    const HWVar& h = hw[0];
    const int q = jq + ((int)threadIdx.x >> g.lg2);
    if (q < nplanes && h.r >= 0) {
      const int cl = mdiv(q, g.magic_PD), pd = q - cl * g.PD;
      const int gd = d0 + pd - g.pd;
      const bool in = h.in && (c0 + cl < g.x.C) && (unsigned)gd < (unsigned)g.x.D;
      if (in && g.tr.scale) {
        const float* tab = reinterpret_cast<const float*>(choff);
        if (g.tr.pre_relu) v = fmaxf(v, 0.f);
        v = v * tab[64 + cl] + tab[128 + cl];
        if (g.tr.post_relu) v = fmaxf(v, 0.f);
      }
      ldsA[cl * g.PSP + pd * g.plane + h.r] = v;
    }
  }
}

template <int U, int J = 0>
__device__ __forceinline__ void patch_issue_u(const PatchDesc& g, const unsigned* choff,
                                              const HWVar (&hw)[U ? U : 1], const crn_rsrc& rs, int nplanes,
                                              int npass, int c0, int d0, float (&val)[PREG]) {
  if constexpr (J < PREG) {
    if (J < npass) patch_issue_one<J, U>(g, choff, hw, rs, nplanes, c0, d0, val[J]);
    patch_issue_u<U, J + 1>(g, choff, hw, rs, nplanes, npass, c0, d0, val);
  }
}
template <int U, int J = 0>
__device__ __forceinline__ void patch_commit_u(const PatchDesc& g, const unsigned* choff,
                                               const HWVar (&hw)[U ? U : 1], float* ldsA, int nplanes, int npass,
                                               int c0, int d0, const float (&val)[PREG]) {
  if constexpr (J < PREG) {
    if (J < npass) patch_commit_one<J, U>(g, choff, hw, ldsA, nplanes, c0, d0, val[J]);
    patch_commit_u<U, J + 1>(g, choff, hw, ldsA, nplanes, npass, c0, d0, val);
  }
}

// Runtime (wave-uniform) dispatch on the plane padding; HWVar storage is sized for the largest U.
struct PatchHW { HWVar v[4]; };
__device__ __forceinline__ void patch_prepare(const PatchDesc& g, int h0, int w0, PatchHW& p) {
  switch (g.lg2) {
    case 8: patch_hw<1>(g, h0, w0, reinterpret_cast<HWVar(&)[1]>(p.v)); break;
    case 9: patch_hw<2>(g, h0, w0, reinterpret_cast<HWVar(&)[2]>(p.v)); break;
    case 10: patch_hw<4>(g, h0, w0, p.v); break;
    default: patch_hw<0>(g, h0, w0, reinterpret_cast<HWVar(&)[1]>(p.v)); break;
  }
}
__device__ __forceinline__ void patch_issue(const PatchDesc& g, const unsigned* choff, const PatchHW& p,
                                            const crn_rsrc& rs, int nplanes, int npass, int c0, int d0,
                                            float (&val)[PREG]) {
  switch (g.lg2) {   // wave-uniform
    case 8: patch_issue_u<1>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), rs, nplanes, npass, c0, d0, val); break;
    case 9: patch_issue_u<2>(g, choff, reinterpret_cast<const HWVar(&)[2]>(p.v), rs, nplanes, npass, c0, d0, val); break;
    case 10: patch_issue_u<4>(g, choff, p.v, rs, nplanes, npass, c0, d0, val); break;
    default: patch_issue_u<0>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), rs, nplanes, npass, c0, d0, val); break;
  }
}
__device__ __forceinline__ void patch_commit(const PatchDesc& g, const unsigned* choff, const PatchHW& p,
                                             float* ldsA, int nplanes, int npass, int c0, int d0,
                                             const float (&val)[PREG]) {
  switch (g.lg2) {
    case 8: patch_commit_u<1>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
    case 9: patch_commit_u<2>(g, choff, reinterpret_cast<const HWVar(&)[2]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
    case 10: patch_commit_u<4>(g, choff, p.v, ldsA, nplanes, npass, c0, d0, val); break;
    default: patch_commit_u<0>(g, choff, reinterpret_cast<const HWVar(&)[1]>(p.v), ldsA, nplanes, npass, c0, d0, val); break;
  }
}

template <int V>
struct IntC { static constexpr int value = V; };

// ------------------------------- forward -----------------------------------
template <int MSUB, int NSUB>
__global__ __launch_bounds__(256, (MSUB * NSUB > 8 ? 1 : 2)) void conv_fwd_kernel(ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);      // 2 x kChTab per-chunk channel tables
  float* ldsA = lds + 2 * kChTab;
  float* ldsB = ldsA + g.CC * g.PSP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16;

  int tile = blockIdx.x;
  const int twi = tile % g.tilesW; tile /= g.tilesW;
  const int thi = tile % g.tilesH; tile /= g.tilesH;
  const int tdi = tile % g.tilesD; tile /= g.tilesD;
  const int b = tile;
  const int d0 = tdi * g.TD, h0 = thi * g.TH, w0 = twi * g.TW;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int cbeg = split * g.chunks_per_split;
  const int cend = min(cbeg + g.chunks_per_split, g.nchunks);

  PatchDesc pdsc;
  pdsc.x = g.x; pdsc.tr = g.tr; pdsc.pd = g.pd; pdsc.ph = g.ph; pdsc.pw = g.pw;
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.plane = g.PH * g.PW; pdsc.lg2 = g.lg2;
  pdsc.PSP = g.PSP; pdsc.magic_PW = g.magic_PW; pdsc.magic_PD = g.magic_PD;
  const int nplanes = g.CC * g.PD, npass = g.npass;
  const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
  PatchHW phw;
  patch_prepare(pdsc, h0, w0, phw);

  // lane's LDS offset of output position (sub-tile s, row i16) at tap (0,0,0)
  int posbase[MSUB];
  const int ri = i16 / g.mw, rj = i16 - ri * g.mw;
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms) {
    int s = wave * MSUB + ms;
    const int sw = s % g.nsw; s /= g.nsw;
    const int sh = s % g.nsh; s /= g.nsh;
    const int sd = s;
    posbase[ms] = (sd * g.PH + sh * g.mh + ri) * g.PW + sw * g.mw + rj + kk * g.PSP;
  }
  const int bbase = kk * g.WSP + i16;

  f32x4 acc[MSUB][NSUB];
#pragma unroll
  for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[ms][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float pval[PREG];
  f32x4 wval[WREG];
  const int nf4 = g.CC * g.T * (NB / 4);

  // weight element f (float4) -> (LDS offset, global float offset or -1)
  auto weight_elem = [&](int f, int c0, int& ldso, int64_t& go) {
    const int j4 = f % (NB / 4);
    const int ct = f / (NB / 4);
    const int cl = mdiv(ct, g.magic_T);
    const int t = ct - cl * g.T;
    const int c = c0 + cl;
    const int n = n0 + j4 * 4;
    ldso = cl * g.WSP + t * NB + j4 * 4;
    go = (c < g.x.C && n < g.Npad) ? ((int64_t)c * g.T + t) * g.Npad + n : -1;
  };
  auto weights_issue = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WREG; ++j) {
      int f = tid + j * 256;
      asm volatile("" : "+v"(f));
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (f < nf4) {
        int ldso; int64_t go;
        weight_elem(f, c0, ldso, go);
        if (go >= 0) v = *reinterpret_cast<const f32x4*>(g.w + go);
      }
      wval[j] = v;
    }
  };
  auto weights_commit = [&](int c0) {
#pragma unroll
    for (int j = 0; j < WREG; ++j) {
      int f = tid + j * 256;
      asm volatile("" : "+v"(f));
      if (f < nf4) {
        int ldso; int64_t go;
        weight_elem(f, c0, ldso, go);
        *reinterpret_cast<f32x4*>(ldsB + ldso) = wval[j];
      }
    }
  };

  if (cbeg < cend) {
    stage_choff(g.x, g.tr, choff + (cbeg & 1) * kChTab, cbeg * g.CC, g.CC);
    __syncthreads();
    patch_issue(pdsc, choff + (cbeg & 1) * kChTab, phw, xrs, nplanes, npass, cbeg * g.CC, d0, pval);
    weights_issue(cbeg * g.CC);
  }
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int c0 = chunk * g.CC;
    __syncthreads();                       // previous chunk's MFMA reads are done
    patch_commit(pdsc, choff + (chunk & 1) * kChTab, phw, ldsA, nplanes, npass, c0, d0, pval);
    weights_commit(c0);
    if (chunk + 1 < cend) stage_choff(g.x, g.tr, choff + ((chunk + 1) & 1) * kChTab, c0 + g.CC, g.CC);
    __syncthreads();
    if (chunk + 1 < cend) {                // next chunk's loads fly under this chunk's MFMAs
      patch_issue(pdsc, choff + ((chunk + 1) & 1) * kChTab, phw, xrs, nplanes, npass, c0 + g.CC, d0, pval);
      weights_issue(c0 + g.CC);
    }

    // MFMA loop.  One (k-step, zd, zh) row of KW taps is straight-line code: the tap offsets
    // along W are ds_read immediates, so a row costs MSUB+1 address adds for KW*MSUB*NSUB MFMAs.
    const int ksteps = g.CC >> 2;
    auto row = [&](auto kwc, int aoff, int boff) {
      constexpr int KW = decltype(kwc)::value;
      const float* pa[MSUB];
#pragma unroll
      for (int ms = 0; ms < MSUB; ++ms) pa[ms] = ldsA + aoff + posbase[ms];
      const float* pb = ldsB + boff + bbase;
#pragma unroll
      for (int zw = 0; zw < KW; ++zw) {
        float a[MSUB], bv[NSUB];
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms) a[ms] = pa[ms][zw];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bv[ns] = pb[zw * NB + ns * 16];
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns)
            acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
      }
    };
    for (int ks = 0; ks < ksteps; ++ks)
      for (int zd = 0; zd < g.kd; ++zd)
        for (int zh = 0; zh < g.kh; ++zh) {
          const int aoff = ks * 4 * g.PSP + (zd * g.PH + zh) * g.PW;
          const int boff = ks * 4 * g.WSP + (zd * g.kh + zh) * g.kw * NB;
          switch (g.kw) {
            case 1: row(IntC<1>{}, aoff, boff); break;
            case 2: row(IntC<2>{}, aoff, boff); break;
            case 3: row(IntC<3>{}, aoff, boff); break;
            case 4: row(IntC<4>{}, aoff, boff); break;
            case 5: row(IntC<5>{}, aoff, boff); break;
            case 7: row(IntC<7>{}, aoff, boff); break;
            default:
              for (int zw = 0; zw < g.kw; ++zw) row(IntC<1>{}, aoff + zw, boff + zw * NB);
          }
        }
  }

  // epilogue: D row = kk*4 + r (position), col = i16 (channel)
  float* yb = g.y.base + (int64_t)b * g.y.sB;
#pragma unroll
  for (int ns = 0; ns < NSUB; ++ns) {
    const int n = n0 + ns * 16 + i16;
    if (n >= g.y.C) continue;
    const int64_t co = view_chan(g.y, n);
    const float bsv = (g.bias && split == 0) ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      int s = wave * MSUB + ms;
      const int sw = s % g.nsw; s /= g.nsw;
      const int sh = s % g.nsh; s /= g.nsh;
      const int sd = s;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row_ = kk * 4 + r;
        const int rr = row_ / g.mw, rc = row_ - rr * g.mw;
        const int od = d0 + sd, oh = h0 + sh * g.mh + rr, ow = w0 + sw * g.mw + rc;
        if (od < g.y.D && oh < g.y.H && ow < g.y.W) {
          float* dst = yb + co + (int64_t)od * g.y.sD + (int64_t)oh * g.y.sH + (int64_t)ow * g.y.sW;
          const float v = acc[ms][ns][r] + bsv;
          if (g.mode == 0) *dst = v;
          else if (g.mode == 1) *dst += v;
          else atomicAdd(dst, v);
        }
      }
    }
  }
}

// ------------------------------ weight grad ---------------------------------
struct WgradGeom {
  crnView x, dy;
  crnInTransform tr;
  float* dw;
  int Npad;
  int kd, kh, kw, pd, ph, pw, T;
  int TD, TH, TW;
  int PD, PH, PW, PSP;
  int lg2, npass;          // patch staging slots
  int dlg2, dnpass;        // dy staging slots (plane = TD*TH*TW positions of one channel)
  int CC;                  // channels per block (rows = CC*T <= 64*RSUB)
  int tilesD, tilesH, tilesW, ntiles;   // ntiles includes batch
  int tiles_per_split;
  unsigned magic_PW, magic_PD, magic_T, magic_TW, magic_TH;
};

template <int RSUB, int NSUB>
__global__ __launch_bounds__(256, (RSUB * NSUB > 8 ? 1 : 2)) void conv_wgrad_kernel(WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);   // x channel table; dy channel offsets at [192,256)
  float* ldsA = lds + 2 * kChTab;           // CC * PSP   (input patch)
  float* ldsB = ldsA + g.CC * g.PSP;        // TD*TH*TW * NBP (dy, [pos][n])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16, NBP = NB + 1;
  const int c0 = blockIdx.x * g.CC;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int nrows = min(g.CC, g.x.C - c0) * g.T;

  PatchDesc pdsc;
  pdsc.x = g.x; pdsc.tr = g.tr; pdsc.pd = g.pd; pdsc.ph = g.ph; pdsc.pw = g.pw;
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.plane = g.PH * g.PW; pdsc.lg2 = g.lg2;
  pdsc.PSP = g.PSP; pdsc.magic_PW = g.magic_PW; pdsc.magic_PD = g.magic_PD;
  const int nplanes = g.CC * g.PD, npass = g.npass;

  // row (c_local, tap) -> LDS offset inside the patch
  int rowbase[RSUB];
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs) {
    int row = (wave * RSUB + rs) * 16 + i16;
    if (row >= nrows) row = 0;                       // never stored
    const int cl = mdiv(row, g.magic_T);
    int t = row - cl * g.T;
    const int zw = t % g.kw; t /= g.kw;
    const int zh = t % g.kh; t /= g.kh;
    rowbase[rs] = cl * g.PSP + (t * g.PH + zh) * g.PW + zw + kk;
  }

  f32x4 acc[RSUB][NSUB];
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) acc[rs][ns] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tbeg = split * g.tiles_per_split;
  const int tend = min(tbeg + g.tiles_per_split, g.ntiles);
  const int npos = g.TD * g.TH * g.TW;

  float pval[PREG];
  float dval[DREG];

  auto tile_origin = [&](int tl, int& b, int& d0, int& h0, int& w0) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    b = tile; d0 = tdi * g.TD; h0 = thi * g.TH; w0 = twi * g.TW;
  };
  // dy tile: planes = channels nl, in-plane index r = position (td,th,tw); LDS layout [pos][n]
  // (lanes run along w: coalesced global reads, odd LDS stride NBP: conflict-free writes).
  // Per-thread position variants (npos <= 512 -> at most 2) are hoisted per tile.
  HWVar dhw0, dhw1;
  auto dy_prepare = [&](int d0, int h0, int w0) {
    auto one = [&](int u, HWVar& out) {
      const int r = g.dlg2 >= 8 ? tid + u * 256 : (tid & ((1 << g.dlg2) - 1));
      const bool valid = r < npos && (u == 0 || g.dlg2 == 9);
      const int r1 = mdiv(r, g.magic_TW), tw = r - r1 * g.TW;
      const int td = mdiv(r1, g.magic_TH), th = r1 - td * g.TH;
      const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
      out.r = valid ? r : -1;
      out.in = valid && od < g.dy.D && oh < g.dy.H && ow < g.dy.W;
      out.off = (unsigned)od * (unsigned)g.dy.sD + (unsigned)oh * (unsigned)g.dy.sH +
                (unsigned)ow * (unsigned)g.dy.sW;
    };
    one(0, dhw0);
    one(1, dhw1);
  };
  // slot J -> (channel q, variant u); q wave-uniform when a plane spans >= 256 slots
  auto dy_issue = [&](const crn_rsrc& rs, auto jc, float& v) {
    constexpr int J = decltype(jc)::value;
    unsigned goff = 0xFFFFFFFFu;
    if (g.dlg2 >= 8) {
      int q = __builtin_amdgcn_readfirstlane(g.dlg2 == 9 ? J / 2 : J);
      asm volatile("" : "+s"(q));
      const HWVar h = (J % 2 == 1 && g.dlg2 == 9) ? dhw1 : dhw0;
      if (q < NB && n0 + q < g.dy.C) goff = h.in ? (chan_off_s(g.dy, n0 + q) + h.off) * 4u : 0xFFFFFFFFu;
    } else {
      int jq = J * (256 >> g.dlg2);
      asm volatile("" : "+s"(jq));
      const int q = jq + (tid >> g.dlg2);
      if (dhw0.in && q < NB && n0 + q < g.dy.C) goff = (choff[kChTab + q] + dhw0.off) * 4u;
    }
    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)goff, 0, 0));
  };
  auto dy_commit = [&](auto jc, float v) {
    constexpr int J = decltype(jc)::value;
    if (g.dlg2 >= 8) {
      int q = __builtin_amdgcn_readfirstlane(g.dlg2 == 9 ? J / 2 : J);
      asm volatile("" : "+s"(q));
      const HWVar h = (J % 2 == 1 && g.dlg2 == 9) ? dhw1 : dhw0;
      if (q < NB && h.r >= 0) ldsB[h.r * NBP + q] = v;
    } else {
      int jq = J * (256 >> g.dlg2);
      asm volatile("" : "+s"(jq));
      const int q = jq + (tid >> g.dlg2);
      if (q < NB && dhw0.r >= 0) ldsB[dhw0.r * NBP + q] = v;
    }
  };
#define CRN_DY_8(OP, A, B0)                                                                        \
  if (g.dnpass > B0) {                                                                             \
    OP(A IntC<B0 + 0>{}, dval[B0 + 0]); OP(A IntC<B0 + 1>{}, dval[B0 + 1]);                        \
    OP(A IntC<B0 + 2>{}, dval[B0 + 2]); OP(A IntC<B0 + 3>{}, dval[B0 + 3]);                        \
    OP(A IntC<B0 + 4>{}, dval[B0 + 4]); OP(A IntC<B0 + 5>{}, dval[B0 + 5]);                        \
    OP(A IntC<B0 + 6>{}, dval[B0 + 6]); OP(A IntC<B0 + 7>{}, dval[B0 + 7]);                        \
  }
#define CRN_DY_ISSUE(rs) do { CRN_DY_8(dy_issue, rs CRN_COMMA, 0) CRN_DY_8(dy_issue, rs CRN_COMMA, 8) \
                              CRN_DY_8(dy_issue, rs CRN_COMMA, 16) CRN_DY_8(dy_issue, rs CRN_COMMA, 24) } while (0)
#define CRN_DY_COMMIT() do { CRN_DY_8(dy_commit, , 0) CRN_DY_8(dy_commit, , 8) CRN_DY_8(dy_commit, , 16) \
                             CRN_DY_8(dy_commit, , 24) } while (0)
#define CRN_COMMA ,

  int cb = 0, cd0 = 0, ch0 = 0, cw0 = 0;     // origin of the tile currently held in registers
  PatchHW phw;
  stage_choff(g.x, g.tr, choff, c0, g.CC);
  if (tid < NB) {
    const int n = min(n0 + tid, g.dy.C - 1);
    choff[kChTab + tid] = g.dy.chan_off ? (unsigned)g.dy.chan_off[n] : (unsigned)n * (unsigned)g.dy.sC;
  }
  __syncthreads();
  if (tbeg < tend) {
    tile_origin(tbeg, cb, cd0, ch0, cw0);
    const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)cb * g.x.sB);
    const crn_rsrc drs = make_rsrc(g.dy.base + (int64_t)cb * g.dy.sB);
    patch_prepare(pdsc, ch0, cw0, phw);
    dy_prepare(cd0, ch0, cw0);
    patch_issue(pdsc, choff, phw, xrs, nplanes, npass, c0, cd0, pval);
    CRN_DY_ISSUE(drs);
  }
  for (int tl = tbeg; tl < tend; ++tl) {
    __syncthreads();
    patch_commit(pdsc, choff, phw, ldsA, nplanes, npass, c0, cd0, pval);
    CRN_DY_COMMIT();
    __syncthreads();
    if (tl + 1 < tend) {
      tile_origin(tl + 1, cb, cd0, ch0, cw0);
      const crn_rsrc xrs = make_rsrc(g.x.base + (int64_t)cb * g.x.sB);
      const crn_rsrc drs = make_rsrc(g.dy.base + (int64_t)cb * g.dy.sB);
      patch_prepare(pdsc, ch0, cw0, phw);
      dy_prepare(cd0, ch0, cw0);
      patch_issue(pdsc, choff, phw, xrs, nplanes, npass, c0, cd0, pval);
      CRN_DY_ISSUE(drs);
    }
    // reduction over the tile's positions: one (td,th) row of TW/4 k-steps is straight-line code
    auto row = [&](auto wsc, int aoff, int boff) {
      constexpr int WS = decltype(wsc)::value;
      const float* pa[RSUB];
#pragma unroll
      for (int rs = 0; rs < RSUB; ++rs) pa[rs] = ldsA + aoff + rowbase[rs];
      const float* pb = ldsB + boff;
#pragma unroll
      for (int ws = 0; ws < WS; ++ws) {
        float a[RSUB], bv[NSUB];
#pragma unroll
        for (int rs = 0; rs < RSUB; ++rs) a[rs] = pa[rs][ws * 4];
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bv[ns] = pb[ws * 4 * NBP + ns * 16];
#pragma unroll
        for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns)
            acc[rs][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rs], bv[ns], acc[rs][ns], 0, 0, 0);
      }
    };
    for (int td = 0; td < g.TD; ++td)
      for (int th = 0; th < g.TH; ++th) {
        const int aoff = (td * g.PH + th) * g.PW;
        const int boff = ((td * g.TH + th) * g.TW + kk) * NBP + i16;
        switch (g.TW >> 2) {
          case 1: row(IntC<1>{}, aoff, boff); break;
          case 2: row(IntC<2>{}, aoff, boff); break;
          case 3: row(IntC<3>{}, aoff, boff); break;
          default: row(IntC<4>{}, aoff, boff); break;
        }
      }
  }
#undef CRN_DY_8
#undef CRN_DY_ISSUE
#undef CRN_DY_COMMIT
#undef CRN_COMMA

  // D row = kk*4 + r -> weight row (c_local*T + tap); col = i16 -> n
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row_ = (wave * RSUB + rs) * 16 + kk * 4 + r;
      if (row_ >= nrows) continue;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const int n = n0 + ns * 16 + i16;
        if (n < g.Npad)
          atomicAdd(g.dw + ((int64_t)c0 * g.T + row_) * g.Npad + n, acc[rs][ns][r]);
      }
    }
}

// --------------------------- pointwise (1x1x1) convolutions ---------------------------------
// Y[b][n][m] = bias[n] + sum_c W[c][n] * T(X[b][c][m]) for plain views with contiguous positions:
// a GEMM whose A operand is already K-major in memory (NCHW), so global loads are float4 rows
// and no index arithmetic is needed.  Block = 64 positions x 32 channels, K chunks of 32 staged
// through LDS with register prefetch; wave w owns positions [16w, 16w+16) x 32 channels.
// Used for ResNet 1x1 convs and their data gradients (resnet50.py:62-69,95-107) and the skip
// compress convs (ray_traced_skip_connection.py:38).
struct PwGeom {
  const float* x; float* y; const float* w; const float* bias;
  crnInTransform tr;
  int B, C, N, Npad, S;            // S = positions per sample
  int64_t xsB, ysB;                // batch strides; channel stride = S for both
  int bias_sB, mode;
};

__global__ __launch_bounds__(256) void pointwise_fwd_kernel(PwGeom g) {
  constexpr int BM = 64, BN = 32, KC = 32, SA = BM + 16, SB = BN + 16;
  __shared__ __attribute__((aligned(16))) float ldsA[KC * SA];
  __shared__ __attribute__((aligned(16))) float ldsB[KC * SB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, b = blockIdx.z;
  const float* xb = g.x + (int64_t)b * g.xsB;
  // staging roles: A: 2 float4 per thread (rows ka0, ka0+16), B: 1 float4 per thread
  const int ka = tid >> 4, ca = (tid & 15) * 4;       // A row / column
  const int kb = tid >> 3, cb = (tid & 7) * 4;        // B row / column
  const bool a_ok = (m0 + ca) < g.S;                  // S % 4 == 0 (checked on the host)
  const bool b_ok = (n0 + cb) < g.Npad;
  f32x4 ra0, ra1, rb;
  auto issue = [&](int c0) {
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c_a0 = c0 + ka, c_a1 = c0 + ka + 16, c_b = c0 + kb;
    ra0 = (a_ok && c_a0 < g.C) ? *reinterpret_cast<const f32x4*>(xb + (int64_t)c_a0 * g.S + m0 + ca) : z;
    ra1 = (a_ok && c_a1 < g.C) ? *reinterpret_cast<const f32x4*>(xb + (int64_t)c_a1 * g.S + m0 + ca) : z;
    rb = (b_ok && c_b < g.C) ? *reinterpret_cast<const f32x4*>(g.w + (int64_t)c_b * g.Npad + n0 + cb) : z;
  };
  auto xform = [&](f32x4 v, int c) -> f32x4 {
    if (g.tr.scale && a_ok && c < g.C) {
      const float sc = g.tr.scale[c], sh = g.tr.shift[c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float t = v[i];
        if (g.tr.pre_relu) t = fmaxf(t, 0.f);
        t = t * sc + sh;
        if (g.tr.post_relu) t = fmaxf(t, 0.f);
        v[i] = t;
      }
    }
    return v;
  };
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  issue(0);
  for (int c0 = 0; c0 < g.C; c0 += KC) {
    __syncthreads();
    *reinterpret_cast<f32x4*>(ldsA + ka * SA + ca) = xform(ra0, c0 + ka);
    *reinterpret_cast<f32x4*>(ldsA + (ka + 16) * SA + ca) = xform(ra1, c0 + ka + 16);
    *reinterpret_cast<f32x4*>(ldsB + kb * SB + cb) = rb;
    __syncthreads();
    if (c0 + KC < g.C) issue(c0 + KC);
    const float* pa = ldsA + kk * SA + wave * 16 + i16;
    const float* pb = ldsB + kk * SB + i16;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const float a = pa[ks * 4 * SA];
      const float b0 = pb[ks * 4 * SB], b1 = pb[ks * 4 * SB + 16];
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[1], 0, 0, 0);
    }
  }
  // D: rows kk*4..kk*4+3 = 4 consecutive positions, col i16 = channel -> one float4 per lane
  const int m = m0 + wave * 16 + kk * 4;
  if (m < g.S) {
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
      const int n = n0 + ns * 16 + i16;
      if (n < g.N) {
        const float bsv = g.bias ? g.bias[(int64_t)b * g.bias_sB + n] : 0.f;
        float* dst = g.y + (int64_t)b * g.ysB + (int64_t)n * g.S + m;
        f32x4 v = acc[ns] + bsv;
        if (g.mode == 1) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      }
    }
  }
}

bool plain_view(const crnView& v) {
  return v.chan_off == nullptr && v.sW == 1 && v.sH == v.W && (v.D == 1 || v.sD == v.H * v.W) &&
         v.sC == (int64_t)v.D * v.H * v.W && (((uintptr_t)v.base) & 15) == 0 && (v.sB & 3) == 0;
}

__global__ void zero_view_kernel(crnView v) {
  const int64_t per_b = (int64_t)v.C * v.D * v.H * v.W;
  const int64_t total = per_b * v.B;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = e;
    const int w = r % v.W; r /= v.W;
    const int h = r % v.H; r /= v.H;
    const int d = r % v.D; r /= v.D;
    const int c = r % v.C; r /= v.C;
    const int64_t co = v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
    v.base[r * v.sB + co + (int64_t)d * v.sD + (int64_t)h * v.sH + (int64_t)w * v.sW] = 0.f;
  }
}

int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
unsigned magic20(int d) { return (unsigned)(((1u << 20) + d - 1) / d); }
// staging slots (planes padded to a power of two) needed for nplanes planes of `plane` elements
int stage_passes(int nplanes, int plane) {
  if (plane > 1024) return 1 << 30;     // staging handles planes padded to at most 1024 slots
  return crn_cdiv((int64_t)nplanes << ilog2_ceil(plane), 256);
}

int pad16mod32(int v) {  // smallest v' >= v with v' % 32 == 16
  int r = v % 32;
  return r <= 16 ? v + (16 - r) : v + (48 - r);
}

constexpr size_t kLdsBudget = 72 * 1024;   // 2 workgroups per CU (160 KiB LDS)

// ---------------- forward configuration search -------------------------------
struct FwdCfg {
  int MSUB, NSUB, CC, mw, mh, tsd, tsh, tsw;
  int64_t blocks;
  size_t lds;
};

// Sub-tile = 1 x mh x mw output positions (16 MFMA rows); tile = tsd x tsh x tsw sub-tiles.
bool fwd_cfg(int MSUB, int NSUB, int B, int Cin, int Npad, int D, int H, int W, int kd, int kh, int kw,
             FwdCfg* out) {
  FwdCfg c;
  c.MSUB = MSUB; c.NSUB = NSUB;
  c.mw = W >= 16 ? 16 : (W >= 8 ? 8 : (W >= 4 ? 4 : (W >= 2 ? 2 : 1)));
  c.mh = 16 / c.mw;
  const int want = 4 * MSUB;
  const int T = kd * kh * kw;
  double best = 1e300;
  c.tsd = 1; c.tsh = 1; c.tsw = want;
  for (int a = 1; a <= want; a *= 2)
    for (int bq = 1; a * bq <= want; bq *= 2) {
      const int cw = want / (a * bq);
      const int TD = a, TH = bq * c.mh, TW = cw * c.mw;
      const double tiles = (double)crn_cdiv(D, TD) * crn_cdiv(H, TH) * crn_cdiv(W, TW);
      const double patch = (double)(TD + kd - 1) * (TH + kh - 1) * (TW + kw - 1);
      const double cost = tiles * (patch + 0.25 * TD * TH * TW);   // loads + wasted MFMA rows
      if (cost < best) { best = cost; c.tsd = a; c.tsh = bq; c.tsw = cw; }
    }
  const int TD = c.tsd, TH = c.tsh * c.mh, TW = c.tsw * c.mw;
  const int PDp = TD + kd - 1, plane = (TH + kh - 1) * (TW + kw - 1);
  const int PS = PDp * plane;
  const int PSP = pad16mod32(PS), WSP = pad16mod32(T * NSUB * 16);
  auto fits = [&](int cc) {
    return (size_t)cc * (PSP + WSP) * 4 + 2 * kChTab * 4 <= kLdsBudget &&
           stage_passes(cc * PDp, plane) <= PREG && (int64_t)cc * T * NSUB * 16 <= 256 * WREG * 4;
  };
  if (!fits(4)) return false;
  int CC = 4;
  const int cin4 = (Cin + 3) & ~3;
  while (CC * 2 <= cin4 && CC * 2 <= 64 && fits(CC * 2)) CC *= 2;
  c.CC = CC;
  c.lds = (size_t)CC * (PSP + WSP) * 4 + 2 * kChTab * 4;
  c.blocks = (int64_t)B * crn_cdiv(D, TD) * crn_cdiv(H, TH) * crn_cdiv(W, TW) * crn_cdiv(Npad, NSUB * 16);
  *out = c;
  return true;
}

template <int MSUB, int NSUB>
int launch_fwd(const ConvGeom& g, dim3 grid, size_t lds_bytes, hipStream_t st) {
  auto k = conv_fwd_kernel<MSUB, NSUB>;
  if (lds_bytes > 65536)
    CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

template <int RSUB, int NSUB>
int launch_wgrad(const WgradGeom& g, dim3 grid, size_t lds_bytes, hipStream_t st) {
  auto k = conv_wgrad_kernel<RSUB, NSUB>;
  if (lds_bytes > 65536)
    CRN_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(k, grid, dim3(256), lds_bytes, st, g);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

}  // namespace

extern "C" int crn_conv_fwd(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                            const float* bias, int bias_sB, const crnView* y,
                            int kd, int kh, int kw, int pd, int ph, int pw,
                            int splits, int accumulate, crnStream stream) {
  if (!x || !y || !w || Npad <= 0 || (Npad & 15) || x->B != y->B || kd < 1 || kh < 1 || kw < 1)
    return CRN_EINVAL;
  if (y->C > Npad) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Sx = (int64_t)x->D * x->H * x->W;
  if (kd * kh * kw == 1 && pd == 0 && ph == 0 && pw == 0 && splits <= 1 && plain_view(*x) && plain_view(*y) &&
      Sx == (int64_t)y->D * y->H * y->W && (Sx & 3) == 0 && (((uintptr_t)w) & 15) == 0) {
    PwGeom p{};
    p.x = x->base; p.y = y->base; p.w = w; p.bias = bias;
    p.tr = tr ? *tr : crnInTransform{nullptr, nullptr, 0, 0};
    p.B = x->B; p.C = x->C; p.N = y->C; p.Npad = Npad; p.S = (int)Sx;
    p.xsB = x->sB; p.ysB = y->sB; p.bias_sB = bias_sB; p.mode = accumulate ? 1 : 0;
    dim3 grid((unsigned)crn_cdiv(Sx, 64), (unsigned)crn_cdiv(y->C, 32), (unsigned)x->B);
    hipLaunchKernelGGL(pointwise_fwd_kernel, grid, dim3(256), 0, st, p);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  // Score every (MSUB, NSUB) tile: useful MFMA rows x operand reuse of the tile x how well
  // the grid (with split-K as a fallback) fills 256 CUs.
  static const int kM[4] = {8, 4, 2, 1};
  static const int kN[3] = {4, 2, 1};
  FwdCfg best{}; bool have = false; double best_score = -1.0;
  for (int mi = 0; mi < 4; ++mi)
    for (int ni = 0; ni < 3; ++ni) {
      if (kN[ni] > 1 && kN[ni] * 16 > Npad) continue;
      if (kM[mi] * kN[ni] > 16) continue;
      FwdCfg c;
      if (!fwd_cfg(kM[mi], kN[ni], y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, &c)) continue;
      const int area = kM[mi] * kN[ni];
      const double reuse = area >= 16 ? 0.85 : area >= 8 ? 0.82 : area >= 4 ? 0.66 : area >= 2 ? 0.5 : 0.4;
      const double npos_tiles = (double)c.blocks / crn_cdiv(Npad, kN[ni] * 16) * (64.0 * kM[mi]);
      const double useful = ((double)y->B * y->D * y->H * y->W) / npos_tiles *
                            ((double)Npad / (crn_cdiv(Npad, kN[ni] * 16) * kN[ni] * 16.0));
      const int nchunks = crn_cdiv(x->C, c.CC);
      double fill = std::min(1.0, (double)c.blocks / 384.0);
      if (c.blocks < 192) fill = 0.8 * std::min(1.0, (double)c.blocks * std::min(nchunks, 16) / 384.0);
      const double score = useful * reuse * fill;
      if (score > best_score) { best_score = score; best = c; have = true; }
    }
  if (const char* f = getenv("CRN_FWD_FORCE")) {     // tuning aid: "MSUB,NSUB"
    int fM, fN; FwdCfg c;
    if (sscanf(f, "%d,%d", &fM, &fN) == 2 && fwd_cfg(fM, fN, y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, &c)) {
      best = c; have = true;
    }
  }
  if (!have) return CRN_EINVAL;
  ConvGeom g{};
  g.x = *x; g.y = *y;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.w = w; g.bias = bias; g.Npad = Npad; g.bias_sB = bias_sB;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.T = kd * kh * kw;
  g.mw = best.mw; g.mh = best.mh; g.nsh = best.tsh; g.nsw = best.tsw;
  g.TD = best.tsd; g.TH = best.tsh * best.mh; g.TW = best.tsw * best.mw;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = g.TW + kw - 1;
  g.PS = g.PD * g.PH * g.PW; g.PSP = pad16mod32(g.PS);
  g.tilesD = crn_cdiv(y->D, g.TD); g.tilesH = crn_cdiv(y->H, g.TH); g.tilesW = crn_cdiv(y->W, g.TW);
  const int NSUB = best.NSUB, CC = best.CC;
  g.CC = CC; g.WSP = pad16mod32(g.T * NSUB * 16);
  g.nchunks = crn_cdiv(x->C, CC);
  if (splits < 1) {   // auto split-K (atomic accumulate) only when the output grid cannot fill the chip
    splits = best.blocks >= 192 ? 1 : (int)std::min<int64_t>(std::min(g.nchunks, 16), crn_cdiv(256, best.blocks));
  }
  if (splits > g.nchunks) splits = g.nchunks;
  g.chunks_per_split = crn_cdiv(g.nchunks, splits);
  splits = crn_cdiv(g.nchunks, g.chunks_per_split);
  g.mode = splits > 1 ? 2 : (accumulate ? 1 : 0);
  g.lg2 = ilog2_ceil(g.PH * g.PW); g.npass = stage_passes(CC * g.PD, g.PH * g.PW);
  g.magic_PW = magic20(g.PW); g.magic_PD = magic20(g.PD); g.magic_T = magic20(g.T);
  if (g.mode == 2 && !accumulate) {
    const int64_t tot = (int64_t)y->B * y->C * y->D * y->H * y->W;
    hipLaunchKernelGGL(zero_view_kernel, dim3((unsigned)std::min<int64_t>(crn_cdiv(tot, 256), 4096)),
                       dim3(256), 0, st, *y);
    CRN_CHECK_LAUNCH();
  }
  dim3 grid((unsigned)(g.tilesD * g.tilesH * g.tilesW * y->B), (unsigned)crn_cdiv(Npad, NSUB * 16),
            (unsigned)splits);
  const size_t lds_bytes = best.lds;
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_fwd] x(C%d %dx%dx%d) y(C%d %dx%dx%d) k%dx%dx%d: MSUB %d NSUB %d CC %d tile %dx%dx%d "
            "grid %ux%ux%u lds %zu npass %d lg2 %d\n", x->C, x->D, x->H, x->W, y->C, y->D, y->H, y->W, kd, kh, kw,
            best.MSUB, NSUB, CC, g.TD, g.TH, g.TW, grid.x, grid.y, grid.z, lds_bytes, g.npass, g.lg2);
#define CRN_FWD_CASE(M, N) if (best.MSUB == M && NSUB == N) return launch_fwd<M, N>(g, grid, lds_bytes, st);
  CRN_FWD_CASE(8, 1) CRN_FWD_CASE(8, 2)
  CRN_FWD_CASE(4, 1) CRN_FWD_CASE(4, 2) CRN_FWD_CASE(4, 4)
  CRN_FWD_CASE(2, 1) CRN_FWD_CASE(2, 2) CRN_FWD_CASE(2, 4)
  CRN_FWD_CASE(1, 1) CRN_FWD_CASE(1, 2) CRN_FWD_CASE(1, 4)
#undef CRN_FWD_CASE
  return CRN_EINVAL;
}

extern "C" int crn_conv_wgrad(const crnView* x, const crnInTransform* tr, const crnView* dy,
                              float* dw, int Npad, int kd, int kh, int kw, int pd, int ph, int pw,
                              int zero_first, crnStream stream) {
  if (!x || !dy || !dw || Npad <= 0 || (Npad & 15) || x->B != dy->B) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T = kd * kh * kw;
  if (T > 512) return CRN_EINVAL;
  const int Dy = dy->D, Hy = dy->H, Wy = dy->W;
  // search (position tile, NSUB, RSUB): minimise estimated cycles per useful MAC
  const int TWc = Wy >= 16 ? 16 : ((Wy + 3) & ~3);
  struct Cand { int TD, TH, TW, RSUB, NSUB, CC; double cost; size_t lds; } best{};
  bool have = false;
  static const int kTD[3] = {4, 2, 1};
  static const int kTH[5] = {16, 8, 4, 2, 1};
  int lastTD = -1;
  for (int tdi = 0; tdi < 3; ++tdi) {
    const int TD = std::min(kTD[tdi], Dy);
    if (TD == lastTD) continue;
    lastTD = TD;
    int lastTH = -1;
    for (int thi = 0; thi < 5; ++thi) {
      const int TH = std::min(kTH[thi], Hy), TW = TWc;
      if (TH == lastTH) continue;
      lastTH = TH;
      const int npos = TD * TH * TW;
      if (npos > 512) continue;
      const int PDp = TD + kd - 1, plane = (TH + kh - 1) * (TW + kw - 1);
      const int PS = PDp * plane, PSP = PS + 1;
      for (int NSUB = 4; NSUB >= 1; NSUB >>= 1) {
        if (NSUB > 1 && NSUB * 16 > Npad) continue;
        const int NB = NSUB * 16;
        if (stage_passes(NB, npos) > DREG) continue;
        for (int RSUB = 8; RSUB >= 1; RSUB >>= 1) {
          if (RSUB * NSUB > 16) continue;
          int CC = std::min(std::min(std::max(1, (64 * RSUB) / T), (int)x->C), 64);
          while (CC > 1 && stage_passes(CC * PDp, plane) > PREG) --CC;
          const size_t lds = (size_t)CC * PSP * 4 + (size_t)npos * (NB + 1) * 4 + 2 * kChTab * 4;
          if (lds > kLdsBudget || stage_passes(CC * PDp, plane) > PREG) continue;
          const int rows = CC * T;
          if (rows > 64 * RSUB) continue;
          if (RSUB > 1 && rows <= 32 * RSUB) continue;  // a smaller RSUB covers it
          const double mfma = (double)RSUB * NSUB * (npos / 4.0) * 32.0;          // cycles per wave
          const double load = ((double)CC * PS + (double)NB * npos) * 4.0 / 6.0;  // ~6 B/clk/CU effective
          const double useful = (double)rows * std::min(NB, Npad) * npos;
          const double cost = (std::max(mfma, load) + 0.25 * std::min(mfma, load) + 1500.0) / useful;
          if (!have || cost < best.cost) { best = Cand{TD, TH, TW, RSUB, NSUB, CC, cost, lds}; have = true; }
        }
      }
    }
  }
  if (const char* f = getenv("CRN_WG_FORCE")) {      // tuning aid: "TD,TH,RSUB,NSUB"
    int fTD, fTH, fR, fN;
    if (sscanf(f, "%d,%d,%d,%d", &fTD, &fTH, &fR, &fN) == 4) {
      fTD = std::min(fTD, Dy); fTH = std::min(fTH, Hy);
      const int PDp = fTD + kd - 1, plane = (fTH + kh - 1) * (TWc + kw - 1);
      int CC = std::min(std::min(std::max(1, (64 * fR) / T), (int)x->C), 64);
      while (CC > 1 && stage_passes(CC * PDp, plane) > PREG) --CC;
      const size_t lds = (size_t)CC * (PDp * plane + 1) * 4 + (size_t)fTD * fTH * TWc * (fN * 16 + 1) * 4 + 2 * kChTab * 4;
      if (stage_passes(fN * 16, fTD * fTH * TWc) <= DREG && CC * T <= 64 * fR && lds <= 150 * 1024) {
        best = Cand{fTD, fTH, TWc, fR, fN, CC, 0.0, lds}; have = true;
      }
    }
  }
  else if (Dy >= 4 && Wy >= 16) {
    // 3-D decoder layers: measured sweep on MI355X (tools/sweep_wgrad.sh): k5 convs run best with the
    // 4x8x16 tile and 512 (channel,tap) rows; the window-4 transposed-conv geometry with a flat 4x2x16
    // tile (small halo, plane <= 128 slots).
    const int fTD = 4, fTH = T >= 100 ? std::min(8, Hy) : std::min(2, Hy);
    const int fN = (T < 100 && Npad >= 32) ? 2 : 1, fR = (T < 100 && Npad >= 32) ? 4 : 8;
    const int PDp = fTD + kd - 1, plane = (fTH + kh - 1) * (TWc + kw - 1);
    int CC = std::min(std::min(std::max(1, (64 * fR) / T), (int)x->C), 64);
    while (CC > 1 && stage_passes(CC * PDp, plane) > PREG) --CC;
    const size_t lds = (size_t)CC * (PDp * plane + 1) * 4 + (size_t)fTD * fTH * TWc * (fN * 16 + 1) * 4 + 2 * kChTab * 4;
    if (stage_passes(fN * 16, fTD * fTH * TWc) <= DREG && CC * T <= 64 * fR && lds <= kLdsBudget) {
      best = Cand{fTD, fTH, TWc, fR, fN, CC, 0.0, lds}; have = true;
    }
  }
  if (!have) return CRN_EINVAL;
  WgradGeom g{};
  g.x = *x; g.dy = *dy;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.dw = dw; g.Npad = Npad;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.T = T;
  g.TD = best.TD; g.TH = best.TH; g.TW = best.TW;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = g.TW + kw - 1;
  g.PSP = g.PD * g.PH * g.PW + 1;
  const int NSUB = best.NSUB, RSUB = best.RSUB, NB = NSUB * 16, CC = best.CC;
  g.CC = CC;
  g.tilesD = crn_cdiv(Dy, g.TD); g.tilesH = crn_cdiv(Hy, g.TH); g.tilesW = crn_cdiv(Wy, g.TW);
  g.ntiles = g.tilesD * g.tilesH * g.tilesW * dy->B;
  const int cblocks = crn_cdiv(x->C, CC), nblocks = crn_cdiv(Npad, NB);
  int splits = std::max(1, std::min(g.ntiles, crn_cdiv(768, cblocks * nblocks)));
  g.tiles_per_split = crn_cdiv(g.ntiles, splits);
  splits = crn_cdiv(g.ntiles, g.tiles_per_split);
  g.lg2 = ilog2_ceil(g.PH * g.PW); g.npass = stage_passes(CC * g.PD, g.PH * g.PW);
  g.dlg2 = ilog2_ceil(g.TD * g.TH * g.TW); g.dnpass = stage_passes(NB, g.TD * g.TH * g.TW);
  g.magic_PW = magic20(g.PW); g.magic_PD = magic20(g.PD); g.magic_T = magic20(T);
  g.magic_TW = magic20(g.TW); g.magic_TH = magic20(g.TH);
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * T * Npad * 4, st));
  dim3 grid((unsigned)cblocks, (unsigned)nblocks, (unsigned)splits);
  const size_t lds_bytes = best.lds;
  static const bool dbg = getenv("CRN_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "[crn_conv_wgrad] x(C%d %dx%dx%d) dy(C%d %dx%dx%d) k%dx%dx%d: RSUB %d NSUB %d CC %d tile %dx%dx%d "
            "grid %ux%ux%u lds %zu npass %d dnpass %d tiles/split %d\n", x->C, x->D, x->H, x->W, dy->C, dy->D, dy->H,
            dy->W, kd, kh, kw, RSUB, NSUB, CC, g.TD, g.TH, g.TW, grid.x, grid.y, grid.z, lds_bytes, g.npass, g.dnpass,
            g.tiles_per_split);
#define CRN_WG_CASE(R, N) if (RSUB == R && NSUB == N) return launch_wgrad<R, N>(g, grid, lds_bytes, st);
  CRN_WG_CASE(8, 1) CRN_WG_CASE(8, 2)
  CRN_WG_CASE(4, 1) CRN_WG_CASE(4, 2) CRN_WG_CASE(4, 4)
  CRN_WG_CASE(2, 1) CRN_WG_CASE(2, 2) CRN_WG_CASE(2, 4)
  CRN_WG_CASE(1, 1) CRN_WG_CASE(1, 2) CRN_WG_CASE(1, 4)
#undef CRN_WG_CASE
  return CRN_EINVAL;
}
