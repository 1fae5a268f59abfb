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
  float inv_PW, inv_PH, inv_PD, inv_T;
};

__device__ __forceinline__ int fdiv(int e, float inv) {  // floor(e / d), exact for e < 2^21
  return (int)(((float)e + 0.5f) * inv);
}

__device__ __forceinline__ int64_t view_chan(const crnView& v, int c) {
  return v.chan_off ? (int64_t)v.chan_off[c] : (int64_t)c * v.sC;
}

// ---- patch staging: global -> registers (issue) and registers -> LDS (commit) ----
struct PatchDesc {
  crnView x;
  crnInTransform tr;
  int pd, ph, pw, PD, PH, PW, PS, PSP;
  float inv_PW, inv_PH, inv_PD;
};

typedef __amdgpu_buffer_rsrc_t crn_rsrc;   // buffer resource (V#)

// Decompose staged element e -> (channel_local, LDS offset, in-bounds, byte offset in the sample).
// Pure function of e: evaluated at issue time for the address and again at commit time for the
// LDS slot, so that only the loaded VALUE lives in registers across the MFMA loop.
struct PatchElem { int cl, lds; bool in; unsigned goff; };
__device__ __forceinline__ PatchElem patch_elem(const PatchDesc& g, const unsigned* choff, int e, int c0,
                                                int d0, int h0, int w0) {
  PatchElem r;
  const int r1 = fdiv(e, g.inv_PW);
  const int pw = e - r1 * g.PW;
  const int r2 = fdiv(r1, g.inv_PH);
  const int ph = r1 - r2 * g.PH;
  r.cl = fdiv(r2, g.inv_PD);
  const int pd = r2 - r.cl * g.PD;
  const int c = c0 + r.cl;
  const int gd = d0 + pd - g.pd, gh = h0 + ph - g.ph, gw = w0 + pw - g.pw;
  r.in = c < g.x.C && (unsigned)gd < (unsigned)g.x.D && (unsigned)gh < (unsigned)g.x.H &&
         (unsigned)gw < (unsigned)g.x.W;
  r.lds = r.cl * g.PSP + (pd * g.PH + ph) * g.PW + pw;
  const unsigned co = choff[r.cl];      // per-chunk channel offsets staged in LDS
  r.goff = r.in ? (co + (unsigned)gd * (unsigned)g.x.sD + (unsigned)gh * (unsigned)g.x.sH +
                   (unsigned)gw * (unsigned)g.x.sW) * 4u
                : 0xFFFFFFFFu;        // out of range of the descriptor -> the load returns 0
  return r;
}

__device__ __forceinline__ crn_rsrc make_rsrc(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFF0u, 0x00020000);
}

// channel offsets (elements) of chunk [c0, c0+nch) -> LDS table (threads 0..nch-1)
__device__ __forceinline__ void stage_choff(const crnView& v, unsigned* choff, int c0, int nch) {
  const int t = threadIdx.x;
  if (t < nch) {
    const int c = min(c0 + t, v.C - 1);
    choff[t] = v.chan_off ? (unsigned)v.chan_off[c] : (unsigned)c * (unsigned)v.sC;
  }
}

__device__ __forceinline__ void patch_issue(const PatchDesc& g, const unsigned* choff, int b, int c0, int d0,
                                            int h0, int w0, int nch, float (&val)[PREG]) {
  const int total = nch * g.PS;
  const crn_rsrc rs = make_rsrc(g.x.base + (int64_t)b * g.x.sB);
#pragma unroll
  for (int j = 0; j < PREG; ++j) {
    int e = threadIdx.x + j * 256;
    asm volatile("" : "+v"(e));          // defeat LICM: recompute per chunk instead of 100+ live VGPRs
    float v = 0.f;
    if (e < total) {
      const PatchElem pe = patch_elem(g, choff, e, c0, d0, h0, w0);
      v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)pe.goff, 0, 0));
    }
    val[j] = v;
    __builtin_amdgcn_sched_barrier(0);   // one address in flight at a time: keeps VGPR pressure flat
  }
}

__device__ __forceinline__ void patch_commit(const PatchDesc& g, const unsigned* choff, float* ldsA, int c0,
                                             int d0, int h0, int w0, int nch, const float (&val)[PREG]) {
  const int total = nch * g.PS;
#pragma unroll
  for (int j = 0; j < PREG; ++j) {
    int e = threadIdx.x + j * 256;
    asm volatile("" : "+v"(e));        // opaque: recompute here, do not keep issue-time values live
    if (e < total) {
      const PatchElem pe = patch_elem(g, choff, e, c0, d0, h0, w0);
      float v = val[j];
      if (pe.in && g.tr.scale) {
        const int c = c0 + pe.cl;
        if (g.tr.pre_relu) v = fmaxf(v, 0.f);
        v = v * g.tr.scale[c] + g.tr.shift[c];
        if (g.tr.post_relu) v = fmaxf(v, 0.f);
      }
      ldsA[pe.lds] = v;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------- forward -----------------------------------
template <int MSUB, int NSUB>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);      // 2 x 64 channel offsets (double buffered)
  float* ldsA = lds + 128;
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
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.PS = g.PS; pdsc.PSP = g.PSP;
  pdsc.inv_PW = g.inv_PW; pdsc.inv_PH = g.inv_PH; pdsc.inv_PD = g.inv_PD;

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
    const int cl = fdiv(ct, g.inv_T);
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
      __builtin_amdgcn_sched_barrier(0);
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
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (cbeg < cend) {
    stage_choff(g.x, choff + (cbeg & 1) * 64, cbeg * g.CC, g.CC);
    __syncthreads();
    patch_issue(pdsc, choff + (cbeg & 1) * 64, b, cbeg * g.CC, d0, h0, w0, g.CC, pval);
    weights_issue(cbeg * g.CC);
  }
  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int c0 = chunk * g.CC;
    __syncthreads();                       // previous chunk's MFMA reads are done
    patch_commit(pdsc, choff + (chunk & 1) * 64, ldsA, c0, d0, h0, w0, g.CC, pval);
    weights_commit(c0);
    if (chunk + 1 < cend) stage_choff(g.x, choff + ((chunk + 1) & 1) * 64, c0 + g.CC, g.CC);
    __syncthreads();
    if (chunk + 1 < cend) {                // next chunk's loads fly under this chunk's MFMAs
      patch_issue(pdsc, choff + ((chunk + 1) & 1) * 64, b, c0 + g.CC, d0, h0, w0, g.CC, pval);
      weights_issue(c0 + g.CC);
    }

    const int ksteps = g.CC >> 2;
    int t = 0;
    for (int zd = 0; zd < g.kd; ++zd)
      for (int zh = 0; zh < g.kh; ++zh)
        for (int zw = 0; zw < g.kw; ++zw, ++t) {
          const int tapoff = (zd * g.PH + zh) * g.PW + zw;
          for (int ks = 0; ks < ksteps; ++ks) {
            float a[MSUB], bv[NSUB];
            const float* pa = ldsA + ks * 4 * g.PSP + tapoff;
            const float* pb = ldsB + ks * 4 * g.WSP + t * NB + bbase;
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms) a[ms] = pa[posbase[ms]];
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) bv[ns] = pb[ns * 16];
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
              for (int ns = 0; ns < NSUB; ++ns)
                acc[ms][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ms], bv[ns], acc[ms][ns], 0, 0, 0);
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
        const int row = kk * 4 + r;
        const int rr = row / g.mw, rc = row - rr * g.mw;
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
  int PD, PH, PW, PS, PSP;
  int NBP;                 // LDS dy row stride (NB + 1)
  int CC;                  // channels per block (rows = CC*T <= 64*RSUB)
  int tilesD, tilesH, tilesW, ntiles;   // ntiles includes batch
  int tiles_per_split;
  float inv_PW, inv_PH, inv_PD, inv_T, inv_TW, inv_TH, inv_TD;
};

template <int RSUB, int NSUB>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned* choff = reinterpret_cast<unsigned*>(lds);   // channel offsets of this block's CC channels
  float* ldsA = lds + 128;                  // CC * PSP   (input patch)
  float* ldsB = ldsA + g.CC * g.PSP;        // TD*TH*TW * NBP (dy, [pos][n])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16;
  const int c0 = blockIdx.x * g.CC;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int nrows = min(g.CC, g.x.C - c0) * g.T;

  PatchDesc pdsc;
  pdsc.x = g.x; pdsc.tr = g.tr; pdsc.pd = g.pd; pdsc.ph = g.ph; pdsc.pw = g.pw;
  pdsc.PD = g.PD; pdsc.PH = g.PH; pdsc.PW = g.PW; pdsc.PS = g.PS; pdsc.PSP = g.PSP;
  pdsc.inv_PW = g.inv_PW; pdsc.inv_PH = g.inv_PH; pdsc.inv_PD = g.inv_PD;

  // row (c_local, tap) -> LDS offset inside the patch
  int rowbase[RSUB];
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs) {
    int row = (wave * RSUB + rs) * 16 + i16;
    if (row >= nrows) row = 0;                       // never stored
    const int cl = fdiv(row, g.inv_T);
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
  const int dytotal = NB * npos;

  float pval[PREG];
  float dval[DREG];

  auto tile_origin = [&](int tl, int& b, int& d0, int& h0, int& w0) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    b = tile; d0 = tdi * g.TD; h0 = thi * g.TH; w0 = twi * g.TW;
  };
  // dy tile -> registers; LDS layout [pos][n] (lanes run along w: coalesced global, odd LDS stride)
  auto dy_elem = [&](int e, int d0, int h0, int w0, int& ldso, unsigned& goff) {
    const int r1 = fdiv(e, g.inv_TW);
    const int tw = e - r1 * g.TW;
    const int r2 = fdiv(r1, g.inv_TH);
    const int th = r1 - r2 * g.TH;
    const int nl = fdiv(r2, g.inv_TD);
    const int td = r2 - nl * g.TD;
    const int n = n0 + nl;
    const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
    const bool in = n < g.dy.C && od < g.dy.D && oh < g.dy.H && ow < g.dy.W;
    ldso = ((td * g.TH + th) * g.TW + tw) * g.NBP + nl;
    const int nn = in ? n : 0;
    const unsigned co = g.dy.chan_off ? (unsigned)g.dy.chan_off[nn] : (unsigned)nn * (unsigned)g.dy.sC;
    goff = in ? (co + (unsigned)od * (unsigned)g.dy.sD + (unsigned)oh * (unsigned)g.dy.sH +
                 (unsigned)ow * (unsigned)g.dy.sW) * 4u
              : 0xFFFFFFFFu;
  };
  auto dy_issue = [&](int b, int d0, int h0, int w0) {
    const crn_rsrc rs = make_rsrc(g.dy.base + (int64_t)b * g.dy.sB);
#pragma unroll
    for (int j = 0; j < DREG; ++j) {
      int e = tid + j * 256;
      asm volatile("" : "+v"(e));
      float v = 0.f;
      if (e < dytotal) {
        int ldso; unsigned goff;
        dy_elem(e, d0, h0, w0, ldso, goff);
        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)goff, 0, 0));
      }
      dval[j] = v;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto dy_commit = [&](int d0, int h0, int w0) {
#pragma unroll
    for (int j = 0; j < DREG; ++j) {
      int e = tid + j * 256;
      asm volatile("" : "+v"(e));
      if (e < dytotal) {
        int ldso; unsigned goff;
        dy_elem(e, d0, h0, w0, ldso, goff);
        ldsB[ldso] = dval[j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  int cb = 0, cd0 = 0, ch0 = 0, cw0 = 0;     // origin of the tile currently held in registers
  stage_choff(g.x, choff, c0, g.CC);
  __syncthreads();
  if (tbeg < tend) {
    tile_origin(tbeg, cb, cd0, ch0, cw0);
    patch_issue(pdsc, choff, cb, c0, cd0, ch0, cw0, g.CC, pval);
    dy_issue(cb, cd0, ch0, cw0);
  }
  for (int tl = tbeg; tl < tend; ++tl) {
    __syncthreads();
    patch_commit(pdsc, choff, ldsA, c0, cd0, ch0, cw0, g.CC, pval);
    dy_commit(cd0, ch0, cw0);
    __syncthreads();
    if (tl + 1 < tend) {
      tile_origin(tl + 1, cb, cd0, ch0, cw0);
      patch_issue(pdsc, choff, cb, c0, cd0, ch0, cw0, g.CC, pval);
      dy_issue(cb, cd0, ch0, cw0);
    }
    // reduction over the tile's positions, 4 consecutive w per MFMA k-step
    for (int td = 0; td < g.TD; ++td)
      for (int th = 0; th < g.TH; ++th) {
        const float* pa = ldsA + (td * g.PH + th) * g.PW;
        const float* pb = ldsB + ((td * g.TH + th) * g.TW + kk) * g.NBP + i16;
        for (int tw = 0; tw < g.TW; tw += 4) {
          float a[RSUB], bv[NSUB];
#pragma unroll
          for (int rs = 0; rs < RSUB; ++rs) a[rs] = pa[rowbase[rs] + tw];
#pragma unroll
          for (int ns = 0; ns < NSUB; ++ns) bv[ns] = pb[tw * g.NBP + ns * 16];
#pragma unroll
          for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
              acc[rs][ns] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rs], bv[ns], acc[rs][ns], 0, 0, 0);
        }
      }
  }

  // D row = kk*4 + r -> weight row (c_local*T + tap); col = i16 -> n
#pragma unroll
  for (int rs = 0; rs < RSUB; ++rs)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (wave * RSUB + rs) * 16 + kk * 4 + r;
      if (row >= nrows) continue;
#pragma unroll
      for (int ns = 0; ns < NSUB; ++ns) {
        const int n = n0 + ns * 16 + i16;
        if (n < g.Npad)
          atomicAdd(g.dw + ((int64_t)c0 * g.T + row) * g.Npad + n, acc[rs][ns][r]);
      }
    }
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
  const int PS = (TD + kd - 1) * (TH + kh - 1) * (TW + kw - 1);
  const int PSP = pad16mod32(PS), WSP = pad16mod32(T * NSUB * 16);
  auto fits = [&](int cc) {
    return (size_t)cc * (PSP + WSP) * 4 + 512 <= kLdsBudget && (int64_t)cc * PS <= kMaxStage &&
           (int64_t)cc * T * NSUB * 16 <= 256 * WREG * 4;
  };
  if (!fits(4)) return false;
  int CC = 4;
  const int cin4 = (Cin + 3) & ~3;
  while (CC * 2 <= cin4 && CC * 2 <= 64 && fits(CC * 2)) CC *= 2;
  c.CC = CC;
  c.lds = (size_t)CC * (PSP + WSP) * 4 + 512;
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
  // Score every (MSUB, NSUB) tile: useful MFMA rows x operand reuse of the tile x how well
  // the grid (with split-K as a fallback) fills 256 CUs.
  static const int kM[4] = {8, 4, 2, 1};
  static const int kN[3] = {4, 2, 1};
  FwdCfg best{}; bool have = false; double best_score = -1.0;
  for (int mi = 0; mi < 4; ++mi)
    for (int ni = 0; ni < 3; ++ni) {
      if (kN[ni] > 1 && kN[ni] * 16 > Npad) continue;
      if (kM[mi] * kN[ni] > 16) continue;            // 128 accumulator VGPRs would spill
      FwdCfg c;
      if (!fwd_cfg(kM[mi], kN[ni], y->B, x->C, Npad, y->D, y->H, y->W, kd, kh, kw, &c)) continue;
      const int area = kM[mi] * kN[ni];
      const double reuse = area >= 32 ? 1.0 : area >= 16 ? 0.92 : area >= 8 ? 0.82 : area >= 4 ? 0.66 : area >= 2 ? 0.5 : 0.4;
      const double npos_tiles = (double)c.blocks / crn_cdiv(Npad, kN[ni] * 16) * (64.0 * kM[mi]);
      const double useful = ((double)y->B * y->D * y->H * y->W) / npos_tiles *
                            ((double)Npad / (crn_cdiv(Npad, kN[ni] * 16) * kN[ni] * 16.0));
      const int nchunks = crn_cdiv(x->C, c.CC);
      double fill = std::min(1.0, (double)c.blocks / 384.0);
      if (c.blocks < 192) fill = 0.8 * std::min(1.0, (double)c.blocks * std::min(nchunks, 16) / 384.0);
      const double score = useful * reuse * fill;
      if (score > best_score) { best_score = score; best = c; have = true; }
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
  g.inv_PW = 1.f / g.PW; g.inv_PH = 1.f / g.PH; g.inv_PD = 1.f / g.PD; g.inv_T = 1.f / g.T;
  if (g.mode == 2 && !accumulate) {
    const int64_t tot = (int64_t)y->B * y->C * y->D * y->H * y->W;
    hipLaunchKernelGGL(zero_view_kernel, dim3((unsigned)std::min<int64_t>(crn_cdiv(tot, 256), 4096)),
                       dim3(256), 0, st, *y);
    CRN_CHECK_LAUNCH();
  }
  dim3 grid((unsigned)(g.tilesD * g.tilesH * g.tilesW * y->B), (unsigned)crn_cdiv(Npad, NSUB * 16),
            (unsigned)splits);
  const size_t lds_bytes = best.lds;
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
      const int PS = (TD + kd - 1) * (TH + kh - 1) * (TW + kw - 1), PSP = PS + 1;
      for (int NSUB = 4; NSUB >= 1; NSUB >>= 1) {
        if (NSUB > 1 && NSUB * 16 > Npad) continue;
        const int NB = NSUB * 16;
        if ((int64_t)NB * npos > 256 * DREG) continue;
        for (int RSUB = 8; RSUB >= 1; RSUB >>= 1) {
          if (RSUB * NSUB > 16) continue;               // accumulator registers
          int CC = std::min(std::min(std::max(1, (64 * RSUB) / T), (int)x->C), 64);
          while (CC > 1 && (int64_t)CC * PS > kMaxStage) --CC;
          const size_t lds = (size_t)CC * PSP * 4 + (size_t)npos * (NB + 1) * 4 + 512;
          if (lds > kLdsBudget || (int64_t)CC * PS > kMaxStage) continue;
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
  if (!have) return CRN_EINVAL;
  WgradGeom g{};
  g.x = *x; g.dy = *dy;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.dw = dw; g.Npad = Npad;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.T = T;
  g.TD = best.TD; g.TH = best.TH; g.TW = best.TW;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = g.TW + kw - 1;
  g.PS = g.PD * g.PH * g.PW; g.PSP = g.PS + 1;
  const int NSUB = best.NSUB, RSUB = best.RSUB, NB = NSUB * 16, CC = best.CC;
  g.NBP = NB + 1; g.CC = CC;
  g.tilesD = crn_cdiv(Dy, g.TD); g.tilesH = crn_cdiv(Hy, g.TH); g.tilesW = crn_cdiv(Wy, g.TW);
  g.ntiles = g.tilesD * g.tilesH * g.tilesW * dy->B;
  const int cblocks = crn_cdiv(x->C, CC), nblocks = crn_cdiv(Npad, NB);
  int splits = std::max(1, std::min(g.ntiles, crn_cdiv(768, cblocks * nblocks)));
  g.tiles_per_split = crn_cdiv(g.ntiles, splits);
  splits = crn_cdiv(g.ntiles, g.tiles_per_split);
  g.inv_PW = 1.f / g.PW; g.inv_PH = 1.f / g.PH; g.inv_PD = 1.f / g.PD; g.inv_T = 1.f / T;
  g.inv_TW = 1.f / g.TW; g.inv_TH = 1.f / g.TH; g.inv_TD = 1.f / g.TD;
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * T * Npad * 4, st));
  dim3 grid((unsigned)cblocks, (unsigned)nblocks, (unsigned)splits);
  const size_t lds_bytes = best.lds;
#define CRN_WG_CASE(R, N) if (RSUB == R && NSUB == N) return launch_wgrad<R, N>(g, grid, lds_bytes, st);
  CRN_WG_CASE(8, 1) CRN_WG_CASE(8, 2)
  CRN_WG_CASE(4, 1) CRN_WG_CASE(4, 2) CRN_WG_CASE(4, 4)
  CRN_WG_CASE(2, 1) CRN_WG_CASE(2, 2) CRN_WG_CASE(2, 4)
  CRN_WG_CASE(1, 1) CRN_WG_CASE(1, 2) CRN_WG_CASE(1, 4)
#undef CRN_WG_CASE
  return CRN_EINVAL;
}
