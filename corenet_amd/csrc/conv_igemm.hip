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
//  * LDS strides are padded to 16 (mod 32) banks so the two 32-lane halves of a
//    ds_read_b32 (k = 0,1 / 2,3) never collide.
//  * BatchRenorm-apply + ReLU of the producer are fused into the patch load
//    (crnInTransform), so normalised activations are never written to HBM.
#include "crn_common.h"
#include <algorithm>

namespace {

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

// Stage CC channels of the input patch for tile origin (b,d0,h0,w0) into LDS.
__device__ __forceinline__ void load_patch(const ConvGeom& g, float* ldsA, int b, int c0,
                                           int d0, int h0, int w0, int nch) {
  const int total = nch * g.PS;
  const float* xb = g.x.base + (int64_t)b * g.x.sB;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int r1 = fdiv(e, g.inv_PW);
    const int pw = e - r1 * g.PW;
    const int r2 = fdiv(r1, g.inv_PH);
    const int ph = r1 - r2 * g.PH;
    const int cl = fdiv(r2, g.inv_PD);
    const int pd = r2 - cl * g.PD;
    const int c = c0 + cl;
    const int gd = d0 + pd - g.pd, gh = h0 + ph - g.ph, gw = w0 + pw - g.pw;
    float v = 0.f;
    if (c < g.x.C && (unsigned)gd < (unsigned)g.x.D && (unsigned)gh < (unsigned)g.x.H &&
        (unsigned)gw < (unsigned)g.x.W) {
      v = xb[view_chan(g.x, c) + (int64_t)gd * g.x.sD + (int64_t)gh * g.x.sH + (int64_t)gw * g.x.sW];
      if (g.tr.scale) {
        if (g.tr.pre_relu) v = fmaxf(v, 0.f);
        v = v * g.tr.scale[c] + g.tr.shift[c];
        if (g.tr.post_relu) v = fmaxf(v, 0.f);
      }
    }
    ldsA[cl * g.PSP + (pd * g.PH + ph) * g.PW + pw] = v;
  }
}

// ------------------------------- forward -----------------------------------
template <int MSUB, int NSUB>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;
  float* ldsB = lds + g.CC * g.PSP;
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

  for (int chunk = cbeg; chunk < cend; ++chunk) {
    const int c0 = chunk * g.CC;
    __syncthreads();
    load_patch(g, ldsA, b, c0, d0, h0, w0, g.CC);
    {  // weight chunk -> LDS [cl][t][NB]
      const int nf4 = g.CC * g.T * (NB / 4);
      for (int f = tid; f < nf4; f += 256) {
        const int j4 = f % (NB / 4);
        const int ct = f / (NB / 4);
        const int cl = fdiv(ct, g.inv_T);
        const int t = ct - cl * g.T;
        const int c = c0 + cl;
        const int n = n0 + j4 * 4;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < g.x.C && n < g.Npad)
          v = *reinterpret_cast<const f32x4*>(g.w + ((int64_t)c * g.T + t) * g.Npad + n);
        *reinterpret_cast<f32x4*>(ldsB + cl * g.WSP + t * NB + j4 * 4) = v;
      }
    }
    __syncthreads();

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
  float inv_PW, inv_PH, inv_PD, inv_T, inv_TW, inv_TH;
};

template <int RSUB, int NSUB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradGeom g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldsA = lds;                        // CC * PSP   (input patch)
  float* ldsB = lds + g.CC * g.PSP;         // TD*TH*TW * NBP (dy, [pos][n])
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, kk = lane >> 4;
  constexpr int NB = NSUB * 16;
  const int c0 = blockIdx.x * g.CC;
  const int n0 = blockIdx.y * NB;
  const int split = blockIdx.z;
  const int nrows = min(g.CC, g.x.C - c0) * g.T;

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
  for (int tl = tbeg; tl < tend; ++tl) {
    int tile = tl;
    const int twi = tile % g.tilesW; tile /= g.tilesW;
    const int thi = tile % g.tilesH; tile /= g.tilesH;
    const int tdi = tile % g.tilesD; tile /= g.tilesD;
    const int b = tile;
    const int d0 = tdi * g.TD, h0 = thi * g.TH, w0 = twi * g.TW;
    __syncthreads();
    {  // patch (same staging as the forward)
      ConvGeom cg;
      cg.x = g.x; cg.tr = g.tr; cg.pd = g.pd; cg.ph = g.ph; cg.pw = g.pw;
      cg.PD = g.PD; cg.PH = g.PH; cg.PW = g.PW; cg.PS = g.PS; cg.PSP = g.PSP;
      cg.inv_PW = g.inv_PW; cg.inv_PH = g.inv_PH; cg.inv_PD = g.inv_PD;
      load_patch(cg, ldsA, b, c0, d0, h0, w0, g.CC);
    }
    {  // dy tile -> LDS [pos][n]  (lanes run along w: coalesced global, odd LDS stride)
      const float* dyb = g.dy.base + (int64_t)b * g.dy.sB;
      const int total = NB * npos;
      for (int e = tid; e < total; e += 256) {
        const int r1 = fdiv(e, g.inv_TW);
        const int tw = e - r1 * g.TW;
        const int r2 = fdiv(r1, g.inv_TH);
        const int th = r1 - r2 * g.TH;
        const int td = r2 % g.TD;
        const int nl = r2 / g.TD;
        const int n = n0 + nl;
        const int od = d0 + td, oh = h0 + th, ow = w0 + tw;
        float v = 0.f;
        if (n < g.dy.C && od < g.dy.D && oh < g.dy.H && ow < g.dy.W)
          v = dyb[view_chan(g.dy, n) + (int64_t)od * g.dy.sD + (int64_t)oh * g.dy.sH + (int64_t)ow * g.dy.sW];
        ldsB[((td * g.TH + th) * g.TW + tw) * g.NBP + nl] = v;
      }
    }
    __syncthreads();
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

struct TileChoice { int MSUB, mw, mh, tsd, tsh, tsw; };

// Sub-tile = 1 x mh x mw output positions (16 MFMA rows); tile = tsd x tsh x tsw sub-tiles.
TileChoice choose_tile(int D, int H, int W, int kd, int kh, int kw) {
  TileChoice tc;
  tc.mw = W >= 16 ? 16 : (W >= 8 ? 8 : (W >= 4 ? 4 : (W >= 2 ? 2 : 1)));
  tc.mh = 16 / tc.mw;
  const int nsw = crn_cdiv(W, tc.mw), nsh = crn_cdiv(H, tc.mh);
  const int64_t total = (int64_t)nsw * nsh * D;
  tc.MSUB = total > 16 ? 8 : (total > 4 ? 4 : 1);
  const int want = 4 * tc.MSUB;
  double best = 1e300;
  tc.tsd = 1; tc.tsh = 1; tc.tsw = want;
  for (int a = 1; a <= want; a *= 2)
    for (int bq = 1; a * bq <= want; bq *= 2) {
      const int c = want / (a * bq);
      // a: along D, bq: along H, c: along W
      const int TD = a, TH = bq * tc.mh, TW = c * tc.mw;
      const double tiles = (double)crn_cdiv(D, TD) * crn_cdiv(H, TH) * crn_cdiv(W, TW);
      const double patch = (double)(TD + kd - 1) * (TH + kh - 1) * (TW + kw - 1);
      const double cost = tiles * (patch + 0.25 * TD * TH * TW);   // loads + wasted MFMA rows
      if (cost < best) { best = cost; tc.tsd = a; tc.tsh = bq; tc.tsw = c; }
    }
  return tc;
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

constexpr size_t kLdsBudget = 64 * 1024;   // 2 workgroups per CU (160 KiB LDS)

}  // namespace

extern "C" int crn_conv_fwd(const crnView* x, const crnInTransform* tr, const float* w, int Npad,
                            const float* bias, int bias_sB, const crnView* y,
                            int kd, int kh, int kw, int pd, int ph, int pw,
                            int splits, int accumulate, crnStream stream) {
  if (!x || !y || !w || Npad <= 0 || (Npad & 15) || x->B != y->B || kd < 1 || kh < 1 || kw < 1)
    return CRN_EINVAL;
  if (y->C > Npad) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  ConvGeom g{};
  g.x = *x; g.y = *y;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.w = w; g.bias = bias; g.Npad = Npad; g.bias_sB = bias_sB;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.T = kd * kh * kw;
  const TileChoice tc = choose_tile(y->D, y->H, y->W, kd, kh, kw);
  g.mw = tc.mw; g.mh = tc.mh; g.nsh = tc.tsh; g.nsw = tc.tsw;
  g.TD = tc.tsd; g.TH = tc.tsh * tc.mh; g.TW = tc.tsw * tc.mw;
  g.PD = g.TD + kd - 1; g.PH = g.TH + kh - 1; g.PW = g.TW + kw - 1;
  g.PS = g.PD * g.PH * g.PW; g.PSP = pad16mod32(g.PS);
  g.tilesD = crn_cdiv(y->D, g.TD); g.tilesH = crn_cdiv(y->H, g.TH); g.tilesW = crn_cdiv(y->W, g.TW);
  // N tile and channel chunk under the LDS budget
  int NSUB = Npad >= 64 ? 4 : (Npad >= 32 ? 2 : 1);
  auto lds_need = [&](int nsub, int cc) {
    return (size_t)cc * ((size_t)g.PSP + pad16mod32(g.T * nsub * 16)) * 4;
  };
  while (NSUB > 1 && lds_need(NSUB, 4) > kLdsBudget) NSUB >>= 1;
  int CC = 4;
  const int cin4 = (x->C + 3) & ~3;
  while (CC * 2 <= cin4 && CC * 2 <= 64 && lds_need(NSUB, CC * 2) <= kLdsBudget) CC *= 2;
  if (lds_need(NSUB, CC) > 160 * 1024) return CRN_EINVAL;
  g.CC = CC; g.WSP = pad16mod32(g.T * NSUB * 16);
  g.nchunks = crn_cdiv(x->C, CC);
  if (splits < 1) {   // auto: fill the 256 CUs (x2 workgroups) when the output grid alone cannot
    const int64_t blocks = (int64_t)g.tilesD * g.tilesH * g.tilesW * y->B * crn_cdiv(Npad, NSUB * 16);
    splits = blocks >= 384 ? 1 : (int)std::min<int64_t>(g.nchunks, crn_cdiv(512, blocks));
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
  const size_t lds_bytes = lds_need(NSUB, CC);
#define CRN_FWD_CASE(M, N) if (tc.MSUB == M && NSUB == N) return launch_fwd<M, N>(g, grid, lds_bytes, st);
  CRN_FWD_CASE(8, 1) CRN_FWD_CASE(8, 2) CRN_FWD_CASE(8, 4)
  CRN_FWD_CASE(4, 1) CRN_FWD_CASE(4, 2) CRN_FWD_CASE(4, 4)
  CRN_FWD_CASE(1, 1) CRN_FWD_CASE(1, 2) CRN_FWD_CASE(1, 4)
#undef CRN_FWD_CASE
  return CRN_EINVAL;
}

extern "C" int crn_conv_wgrad(const crnView* x, const crnInTransform* tr, const crnView* dy,
                              float* dw, int Npad, int kd, int kh, int kw, int pd, int ph, int pw,
                              int zero_first, crnStream stream) {
  if (!x || !dy || !dw || Npad <= 0 || (Npad & 15) || x->B != dy->B) return CRN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  WgradGeom g{};
  g.x = *x; g.dy = *dy;
  if (tr) g.tr = *tr; else g.tr = crnInTransform{nullptr, nullptr, 0, 0};
  g.dw = dw; g.Npad = Npad;
  g.kd = kd; g.kh = kh; g.kw = kw; g.pd = pd; g.ph = ph; g.pw = pw; g.T = kd * kh * kw;
  const int T = g.T;
  // position tile: TW multiple of 4 (k-step = 4 consecutive w), ~256-512 positions
  const int Dy = dy->D, Hy = dy->H, Wy = dy->W;
  int TW = Wy >= 16 ? 16 : ((Wy + 3) & ~3);
  int TH = std::min(Hy, 8), TD = std::min(Dy, 4);
  if (Dy == 1) TH = std::min(Hy, 16);
  g.TD = TD; g.TH = TH; g.TW = TW;
  g.PD = TD + kd - 1; g.PH = TH + kh - 1; g.PW = TW + kw - 1;
  g.PS = g.PD * g.PH * g.PW; g.PSP = g.PS + 1;
  const int NSUB = Npad >= 64 ? 4 : (Npad >= 32 ? 2 : 1);
  const int NB = NSUB * 16;
  g.NBP = NB + 1;
  const size_t dy_bytes = (size_t)TD * TH * TW * g.NBP * 4;
  // rows = CC*T <= 64*RSUB, patch under the LDS budget
  int RSUB = 8;
  int CC = std::max(1, (64 * RSUB) / T);
  CC = std::min(CC, x->C);
  while (CC > 1 && (size_t)CC * g.PSP * 4 + dy_bytes > kLdsBudget + 16384) --CC;
  const int rows = CC * T;
  if (rows > 64 * 8) return CRN_EINVAL;                 // T > 512 unsupported
  RSUB = rows > 256 ? 8 : (rows > 128 ? 4 : (rows > 64 ? 2 : 1));
  g.CC = CC;
  g.tilesD = crn_cdiv(Dy, TD); g.tilesH = crn_cdiv(Hy, TH); g.tilesW = crn_cdiv(Wy, TW);
  g.ntiles = g.tilesD * g.tilesH * g.tilesW * dy->B;
  const int cblocks = crn_cdiv(x->C, CC), nblocks = crn_cdiv(Npad, NB);
  int splits = std::max(1, std::min(g.ntiles, crn_cdiv(1024, cblocks * nblocks)));
  g.tiles_per_split = crn_cdiv(g.ntiles, splits);
  splits = crn_cdiv(g.ntiles, g.tiles_per_split);
  g.inv_PW = 1.f / g.PW; g.inv_PH = 1.f / g.PH; g.inv_PD = 1.f / g.PD; g.inv_T = 1.f / T;
  g.inv_TW = 1.f / TW; g.inv_TH = 1.f / TH;
  if (zero_first) CRN_HIP(hipMemsetAsync(dw, 0, (size_t)x->C * T * Npad * 4, st));
  dim3 grid((unsigned)cblocks, (unsigned)nblocks, (unsigned)splits);
  const size_t lds_bytes = (size_t)CC * g.PSP * 4 + dy_bytes;
  if (lds_bytes > 160 * 1024) return CRN_EINVAL;
#define CRN_WG_CASE(R, N) if (RSUB == R && NSUB == N) return launch_wgrad<R, N>(g, grid, lds_bytes, st);
  CRN_WG_CASE(8, 1) CRN_WG_CASE(8, 2) CRN_WG_CASE(8, 4)
  CRN_WG_CASE(4, 1) CRN_WG_CASE(4, 2) CRN_WG_CASE(4, 4)
  CRN_WG_CASE(2, 1) CRN_WG_CASE(2, 2) CRN_WG_CASE(2, 4)
  CRN_WG_CASE(1, 1) CRN_WG_CASE(1, 2) CRN_WG_CASE(1, 4)
#undef CRN_WG_CASE
  return CRN_EINVAL;
}
