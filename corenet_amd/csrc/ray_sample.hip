// Ray-traced skip connection: per-voxel camera projection + truncating
// nearest-pixel gather (forward) and scatter-add (backward).
// Reference: model/ray_traced_skip_connection.py:91-144 (SampleGrid2d.forward);
// the reference runs ~25 torch kernels, three int64 index tensors of D*H*W and
// a non-contiguous permute; here: one fused kernel, output written straight
// into the decoder's concat buffer, HBM traffic = the algorithmic bytes
// (C*D*H*W*4 written, the C*h*w*4 map read once through L2).
//
// Index arithmetic is bit-defined (oracle/corenet_oracle.py:ray_sample_indices):
// fp32, IEEE round-to-nearest, NO fma contraction, evaluation order
//   c = v + off;  p_n = ((m_n0*cx + m_n1*cy) + m_n2*cz) + m_n3
//   u = (p_x/p_w)/2 + 0.5;  ix = (int)(u*W)   (C truncation toward zero)
#include "crn_common.h"

namespace {

struct Cam { float m[16]; float ox, oy, oz; };

__device__ __forceinline__ Cam load_cam(const float* matrix, const float* offset, int b) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 16; ++i) c.m[i] = matrix[b * 16 + i];
  c.ox = offset[b * 3 + 0]; c.oy = offset[b * 3 + 1]; c.oz = offset[b * 3 + 2];
  return c;
}

__device__ __forceinline__ float row_dot(const float* m, float cx, float cy, float cz) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(m[0], cx), __fmul_rn(m[1], cy)), __fmul_rn(m[2], cz)), m[3]);
}

// flat pixel offset iy*w+ix inside the (unpadded) map, or -1 for "outside value" (0)
__device__ __forceinline__ int project(const Cam& c, int x, int y, int z, int w, int h) {
  const float cx = __fadd_rn((float)x, c.ox), cy = __fadd_rn((float)y, c.oy), cz = __fadd_rn((float)z, c.oz);
  const float px = row_dot(c.m + 0, cx, cy, cz);
  const float py = row_dot(c.m + 4, cx, cy, cz);
  const float pz = row_dot(c.m + 8, cx, cy, cz);
  const float pw = row_dot(c.m + 12, cx, cy, cz);
  // x / 2 == x * 0.5 bit for bit (exact power-of-two scaling, also for subnormals)
  const float u = __fadd_rn(__fmul_rn(__fdiv_rn(px, pw), 0.5f), 0.5f);
  const float v = __fadd_rn(__fmul_rn(__fdiv_rn(py, pw), 0.5f), 0.5f);
  const float fu = __fmul_rn(u, (float)w), fv = __fmul_rn(v, (float)h);
  // (int) cast: truncation; out-of-range values saturate and fall outside [0,w)
  const int ix = (int)fu, iy = (int)fv;
  const bool ok = (pz >= 0.0f) && ix >= 0 && ix < w && iy >= 0 && iy < h &&
                  fu < 2147483520.0f && fv < 2147483520.0f && fu > -2147483520.0f && fv > -2147483520.0f;
  return ok ? iy * w + ix : -1;
}

// forward: one thread = 4 consecutive x voxels and all channels.
// CL = false: map [B][C][h][w]: one scalar gather per (voxel, channel).  Measured at 64^3 x 12 x B=4: stores
// alone 9.9 us, + projection 10.9 us, + the 48 scalar gathers per lane 17.3 us -- a per-lane-address dword
// load costs the texture addresser ~20 cycles per wave whatever it hits, so the gathers, not HBM, set the time.
// CL = true: map [B][h][w][C] (channel-last, written that way by the 1x1 compress conv): the channels of a
// pixel are contiguous, one dwordx4 gather fetches four of them -> 4x fewer gather instructions.
template <int VX, bool CL>
__global__ __launch_bounds__(256) void ray_sample_fwd_kernel(
    const float* __restrict__ map, int64_t map_sB, int64_t map_sC, int64_t map_sP, int C, int h, int w,
    const float* matrix, const float* offset, float* __restrict__ out, int64_t out_sB, int D, int H, int W,
    int64_t per_b, uint16_t* __restrict__ idx) {
  crn_kernargs_now(map, map_sB, map_sC, map_sP, C, h, w, matrix, offset, out, out_sB, D, H, W, per_b, idx);
  const int b = blockIdx.y;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= per_b) return;
  const int wx = W / VX;
  int64_t r = t;
  const int x0 = (int)(r % wx) * VX; r /= wx;
  const int y = (int)(r % H);
  const int z = (int)(r / H);
  const Cam cam = load_cam(matrix, offset, b);
  int po[VX];
#pragma unroll
  for (int k = 0; k < VX; ++k) po[k] = project(cam, x0 + k, y, z, w, h);
  const int64_t S = (int64_t)D * H * W;
  const float* mb = map + (int64_t)b * map_sB;
  float* ob = out + (int64_t)b * out_sB + ((int64_t)z * H + y) * W + x0;
  if (idx && blockIdx.z == 0) {                 // the saved index tensor of the backward pass (crn_ray_sample_bwd_idx)
    uint16_t* ib = idx + (int64_t)b * S + ((int64_t)z * H + y) * W + x0;
    if (VX == 4) {
      typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
      u16x4 v4;
#pragma unroll
      for (int k = 0; k < 4; ++k) v4[k] = (unsigned short)(po[k] >= 0 ? po[k] : 0xFFFF);
      *reinterpret_cast<u16x4*>(ib) = v4;
    } else {
#pragma unroll
      for (int k = 0; k < VX; ++k) ib[k] = (uint16_t)(po[k] >= 0 ? po[k] : 0xFFFF);
    }
  }
  // 12 channels per iteration (every skip width 96/48/24/12 is a multiple): all gathers of the group in
  // flight, then 12 streaming (non-temporal) float4 stores -- the output is consumed much later by the
  // next decoder stage, so it should not displace the feature map from L2.
  // blockIdx.z = one group of 12 channels (the last group also takes the C % 12 leftovers): at the coarse scales
  // (8^3 x 96 ... 32^3 x 24) the groups run side by side instead of one after the other behind store latency
  int c = blockIdx.z * 12;
  const int cend = blockIdx.z + 1 == gridDim.z ? C : c + 12;
  for (; c + 12 <= cend; c += 12) {
    float v[12][VX];
    if (CL) {
#pragma unroll
      for (int k = 0; k < VX; ++k) {
        const float* mp = mb + (int64_t)(po[k] >= 0 ? po[k] : 0) * map_sP + c;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(mp + q * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[q * 4 + i][k] = po[k] >= 0 ? g[i] : 0.f;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const float* mc = mb + (c + u) * map_sC;
#pragma unroll
        for (int k = 0; k < VX; ++k) v[u][k] = po[k] >= 0 ? mc[po[k] * map_sP] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (VX == 4) {
        __builtin_nontemporal_store((f32x4){v[u][0], v[u][1], v[u][2], v[u][3]},
                                    reinterpret_cast<f32x4*>(ob + (c + u) * S));
      } else {
#pragma unroll
        for (int k = 0; k < VX; ++k) ob[(c + u) * S + k] = v[u][k];
      }
    }
  }
  for (; c < cend; ++c) {
    const float* mc = mb + c * map_sC;
    float v[VX];
#pragma unroll
    for (int k = 0; k < VX; ++k) v[k] = po[k] >= 0 ? mc[po[k] * map_sP] : 0.f;
    if (VX == 4) {
      *reinterpret_cast<f32x4*>(ob + c * S) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int k = 0; k < VX; ++k) ob[c * S + k] = v[k];
    }
  }
}

// The saved index tensor of the backward pass (the reference's autograd keeps the int64 index tensors of the forward's
// advanced indexing and scatters with index_put_(accumulate=True), ray_traced_skip_connection.py:124-135): one entry
// per voxel, flat pixel iy*w+ix, kOut<IT>() for "outside value".  16-bit entries when h*w < 65535 (2 B per voxel next
// to the 4*C B of gradient the scatter reads), 32-bit otherwise.
template <typename IT> __device__ __host__ constexpr int idx_out();
template <> __device__ __host__ constexpr int idx_out<uint16_t>() { return 0xFFFF; }
template <> __device__ __host__ constexpr int idx_out<int32_t>() { return -1; }

// idx[b][z][y][x] for every voxel: the projection alone (loop-free: one thread = VX consecutive x voxels, like the gather)
template <int VX, typename IT>
__global__ __launch_bounds__(256) void ray_project_kernel(const float* matrix, const float* offset, IT* __restrict__ idx,
                                                          int D, int H, int W, int h, int w, int64_t per_b) {
  crn_kernargs_now(matrix, offset, idx, D, H, W, h, w, per_b);
  const int b = blockIdx.y;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t >= per_b) return;
  const int wx = W / VX;
  int64_t r = t;
  const int x0 = (int)(r % wx) * VX; r /= wx;
  const int y = (int)(r % H);
  const int z = (int)(r / H);
  const Cam cam = load_cam(matrix, offset, b);
  IT* ib = idx + (int64_t)b * D * H * W + ((int64_t)z * H + y) * W + x0;
#pragma unroll
  for (int k = 0; k < VX; ++k) {
    const int po = project(cam, x0 + k, y, z, w, h);
    ib[k] = (IT)(po >= 0 ? po : idx_out<IT>());
  }
}

// backward (scatter-add of the gradient into the 2-D map) from the saved index tensor.
// A workgroup owns a kTX x kTY tile of (x, y) voxel columns over a z segment of <= ZS planes and CN channels; one thread = one
// column.  Everything a thread needs (ZS indices, ZS x CN gradients) is loaded up front -- no load sits behind a branch and
// the whole segment is in flight at once --, then the column is walked: consecutive z usually hit the same pixel, so runs are
// summed in registers and a run goes into an LDS image of the pixel window the tile reaches; the window goes to HBM with one
// float atomic per touched (pixel, channel).  Same-address float atomics in L2 serialise (measured in round 1: 100 us with
// them, 13 us without), neighbouring columns hit the SAME pixels, hence the window.
// THE WINDOW IS 64-BIT FIXED POINT.  ds_add_f32 executes one lane at a time on this part: 12 cycles per active lane, 768
// cycles for a full wave instruction, whatever the addresses and however many waves wait (tools/lds_atomic_probe.hip,
// profiles/r05_lds_atomic_probe.txt) -- the float window of rounds 1-4 spent 35 of the kernel's 48 us there.  ds_add_u64 is
// 24-33 cycles per wave instruction (48 when pairs of lanes share an address), ds_add_u32 19.  So a run sum v becomes
// F = trunc(v * 2^(44 - e)) with 2^e > M = the largest |gradient| the workgroup loaded (block floating point: every addend is
// resolved to 2^-44 M, a thousand times finer than an fp32 ulp of M, so the window's sum is MORE exact than a chain of float
// adds; |F| < 2^48 and a pixel takes at most 256 x ZS runs, no overflow), integer adds commute (the window's sum does not depend
// on the order of the threads) and the write-out converts once.  A workgroup whose M is inf / NaN adds to HBM directly in
// float (the reference's index_put_ propagates them).
// The window box is the bounding box of the pixels of the tile's 8 corner voxels (a projective map of a box has its extremes
// at the corners as long as pw keeps its sign); it only decides where a run is ADDED, never what is added: a run whose pixel
// lies outside the box (corner voxels outside the image, a box larger than the LDS window, cameras that break the corner
// argument) goes to HBM directly.
// There is no floating-point projection in this kernel (rounds 1-4 projected every voxel again, three times over for the
// three channel groups, and the loop-invariant products of that projection were what the MFMA-neighbour glitch of DESIGN
// section 3e hit); index compares, float adds of run sums and integer atomics are all that is left.
// DET (deterministic mode, crn_common.h): ALL sums are taken in 64-bit fixed point, in HBM -- scale_p[0] = 2^k chosen from
// max |dout| of the whole tensor so that a pixel's sum cannot overflow; detmap = int64 image of dmap ([B][C][h][w], zeroed),
// converted by ray_det_finish_kernel.  The LDS window is not used.
struct ScatterArgs {
  const float* dout; int64_t dout_sB; int C, D, H, W;
  const void* idx;
  float* dmap; int64_t dmap_sB; int h, w;
  unsigned wrecip;                 // ceil(2^32 / w), w >= 2 (the host re-shapes width-1 maps): iy = umulhi(po, wrecip) is exact for po < 2^16
  int zseg, tilesX, tilesY, kTX, kTY, win_floats;      // win_floats: entries (8 bytes each) of the LDS window
  unsigned long long* detmap; const float* scale_p;
  int dbg;                         // tools build only (CRN_RAY_DBG): 1 no LDS adds, 2 no window write-out, 4 every plane = plane z0
};
#ifdef CRN_TOOLS
#define RAY_DBG(a, bit) ((a).dbg & (bit))
#else
#define RAY_DBG(a, bit) 0
#endif

constexpr int kFixBits = 44;        // fixed-point window: a run sum v is added as trunc(v * 2^(kFixBits - e)), M < 2^e

// v * s as a 64-bit integer (|v * s| < 2^55, s a power of two: the product is exact): hi = floor(t / 2^24) and
// lo = t - hi * 2^24 in [0, 2^24) are both exact floats that fit a 32-bit conversion
__device__ __forceinline__ unsigned long long to_fixed(float v, float s) {
  const float t = v * s;
  const float hf = floorf(t * 5.9604644775390625e-08f);                  // 2^-24
  const float lf = __builtin_fmaf(hf, -16777216.f, t);                   // exact
  const long long hi = (long long)(int)hf;
  return (unsigned long long)(hi * 16777216LL + (long long)(unsigned)lf);       // (hi may be negative: no shift of a signed value)
}

template <int CN, int ZS, typename IT, bool DET>
__global__ __launch_bounds__(256) void ray_scatter_kernel(const ScatterArgs a) {
  extern __shared__ unsigned long long win[];
  __shared__ int wbox[4];
  __shared__ float wmax[4];
  crn_kernarg_touch(a);
  const int C = a.C, D = a.D, H = a.H, W = a.W, h = a.h, w = a.w;
  const int b = blockIdx.z;
  const int cbase = blockIdx.y * CN;
  int tile = blockIdx.x;
  const int tx = tile % a.tilesX; tile /= a.tilesX;
  const int ty = tile % a.tilesY; tile /= a.tilesY;
  const int z0 = tile * a.zseg, z1 = min(D, z0 + a.zseg);
  const int x0 = tx * a.kTX, y0 = ty * a.kTY;
  const int x = min(W - 1, x0 + (int)(threadIdx.x % a.kTX)), y = min(H - 1, y0 + (int)(threadIdx.x / a.kTX));
  const bool mine = x0 + (int)(threadIdx.x % a.kTX) < W && y0 + (int)(threadIdx.x / a.kTX) < H;
  const int64_t HW = (int64_t)H * W, S = HW * D, hw = (int64_t)h * w;
  const IT* ib = reinterpret_cast<const IT*>(a.idx) + (int64_t)b * S;
  // every load of the thread, clamped into the tensor: ZS indices, ZS x CN gradients
  int pi[ZS];
  float g[ZS][CN];
  {
    const IT* ip = ib + (int64_t)y * W + x;
    const float* gp = a.dout + (int64_t)b * a.dout_sB + (int64_t)cbase * S + (int64_t)y * W + x;
#pragma unroll
    for (int j = 0; j < ZS; ++j) {
      const int64_t zo = (int64_t)(RAY_DBG(a, 4) ? z0 : min(z0 + j, z1 - 1)) * HW;
      pi[j] = (int)__builtin_nontemporal_load(ip + zo);
#pragma unroll
      for (int k = 0; k < CN; ++k) g[j][k] = __builtin_nontemporal_load(gp + (cbase + k < C ? k : 0) * S + zo);
    }
  }
  // window box from the 8 corner voxels of the tile: lanes 0-7 of the first wave load one corner each
  if (!DET && threadIdx.x < 64) {
    const int k = threadIdx.x & 7;
    const int cx = (k & 1) ? min(W, x0 + a.kTX) - 1 : x0, cy = (k & 2) ? min(H, y0 + a.kTY) - 1 : y0, cz = (k & 4) ? z1 - 1 : z0;
    const int po = (int)ib[((int64_t)cz * H + cy) * W + cx];
    const bool in = po != idx_out<IT>() && po >= 0;
    const int iy = in ? (sizeof(IT) == 2 ? (int)__umulhi((unsigned)po << 0, a.wrecip) : po / w) : 0;
    const int ix = po - iy * w;
    int lox = in ? ix : 0x7fffffff, hix = in ? ix : -1, loy = in ? iy : 0x7fffffff, hiy = in ? iy : -1;
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      lox = min(lox, __shfl_xor(lox, o)); hix = max(hix, __shfl_xor(hix, o));
      loy = min(loy, __shfl_xor(loy, o)); hiy = max(hiy, __shfl_xor(hiy, o));
    }
    if (threadIdx.x == 0) {
      // one pixel of margin: truncation makes the extreme pixel of an interior voxel at most the corners' extreme, the margin is
      // for free (a pixel outside the box is still added, to HBM directly)
      int ix0 = max(0, lox - 1), ix1 = min(w - 1, hix + 1), iy0 = max(0, loy - 1), iy1 = min(h - 1, hiy + 1);
      if (hix < 0 || (int64_t)(ix1 - ix0 + 1) * (iy1 - iy0 + 1) * CN > a.win_floats) { ix0 = iy0 = 0; ix1 = iy1 = -1; }
      wbox[0] = ix0; wbox[1] = ix1 - ix0 + 1; wbox[2] = iy0; wbox[3] = iy1 - iy0 + 1;
    }
  }
  if (DET && threadIdx.x == 0) { wbox[0] = wbox[2] = 0; wbox[1] = wbox[3] = 0; }
  float fix_scale = 0.f, fix_inv = 0.f;
  if (!DET) {
    for (int i = threadIdx.x; i < a.win_floats; i += blockDim.x) win[i] = 0ull;
    // M = the largest |gradient| of the workgroup (NaN-propagating: integer max of the absolute bit patterns)
    unsigned mb = 0;
#pragma unroll
    for (int j = 0; j < ZS; ++j)
#pragma unroll
      for (int k = 0; k < CN; ++k) mb = max(mb, __float_as_uint(g[j][k]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = __uint_as_float(mb);
  }
  __syncthreads();
  const int ix0 = wbox[0], iy0 = wbox[2], wn_box = wbox[1] * wbox[3];
  int ww = wbox[1], wh = wbox[3];
  if (!DET) {
    unsigned mb = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mb = max(mb, __float_as_uint(wmax[i]));
    if (mb == 0) return;                                // every gradient of the workgroup is +-0: nothing to add
    const int E = (int)(mb >> 23);                      // M < 2^(E - 126)
    if (E == 255) { ww = 0; wh = 0; }                   // inf / NaN: no window, float atomics to HBM propagate them
    // scale 2^(kFixBits - (E - 126)): exponent field 127 + kFixBits + 126 - E
    const int ex = 127 + kFixBits + 126 - E;
    fix_scale = __uint_as_float((unsigned)min(ex, 254) << 23);
    fix_inv = __uint_as_float((unsigned)max(254 - min(ex, 254), 1) << 23);     // 1 / fix_scale
    if (ex > 253) { ww = 0; wh = 0; }                   // M < 2^-83: the scale is not a float; such gradients take the float path
  }
  const int wn = wn_box;
  float* mb = a.dmap + (int64_t)b * a.dmap_sB + (int64_t)cbase * hw;
  if (mine) {
    float acc[CN];
#pragma unroll
    for (int k = 0; k < CN; ++k) acc[k] = 0.f;
    int cur = idx_out<IT>();
    auto flush = [&]() {
      if (cur != idx_out<IT>()) {
        const int iy = sizeof(IT) == 2 ? (int)__umulhi((unsigned)cur, a.wrecip) : cur / w;
        const int ix = cur - iy * w;
        const unsigned ux = (unsigned)(ix - ix0), uy = (unsigned)(iy - iy0);
        if (!DET && ux < (unsigned)ww && uy < (unsigned)wh) {
          unsigned long long* wp = win + uy * ww + ux;
          if (RAY_DBG(a, 1)) { if (acc[0] == 123.456f) wp[0] = (unsigned long long)acc[CN - 1]; } else
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(wp + k * wn, to_fixed(acc[k], fix_scale));          // ds_add_u64
        } else if (DET) {
          const float sc = a.scale_p[0];
          unsigned long long* db = a.detmap + ((int64_t)b * C + cbase) * hw;
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(db + k * hw + cur, (unsigned long long)(long long)__float2ll_rn(acc[k] * sc));
        } else {
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(mb + k * hw + cur, acc[k]);
        }
      }
    };
#pragma unroll
    for (int j = 0; j < ZS; ++j) {
      if (z0 + j < z1) {
        if (pi[j] != cur) {
          flush();
          cur = pi[j];
#pragma unroll
          for (int k = 0; k < CN; ++k) acc[k] = g[j][k];
        } else {
#pragma unroll
          for (int k = 0; k < CN; ++k) acc[k] += g[j][k];
        }
      }
    }
    flush();
  }
  if (DET || RAY_DBG(a, 2)) return;
  __syncthreads();
  if (ww == 0) return;
  for (int i = threadIdx.x; i < wn * CN; i += blockDim.x) {
    const long long f = (long long)win[i];
    if (f != 0) {
      const int k = i / wn, p = i - k * wn;
      const int iy = iy0 + p / ww, ix = ix0 + p % ww;
      // one rounding: |f| < 2^60 is exact in double up to 2^53, beyond that the double conversion rounds at 2^-53 relative
      if (cbase + k < C) atomicAdd(mb + k * hw + (int64_t)iy * w + ix, (float)((double)f * (double)fix_inv));
    }
  }
}

// deterministic mode: max |dout| (integer max of the float bit patterns), then the power-of-two scale
__global__ void ray_det_maxabs_kernel(const float* dout, int64_t dout_sB, int64_t per_b, int B, unsigned* maxbits) {
  unsigned m = 0;
  for (int b = 0; b < B; ++b)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_b; i += (int64_t)gridDim.x * blockDim.x)
      m = max(m, __float_as_uint(fabsf(dout[b * dout_sB + i])));
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0 && m) atomicMax(maxbits, m);
}
__global__ void ray_det_scale_kernel(const unsigned* maxbits, float* scale, int log2_terms) {
  // |sum| <= 2^log2_terms * max: scale = 2^(61 - log2_terms - exponent(max) - 1) keeps it below 2^62
  const float mx = __uint_as_float(maxbits[0]);
  int e = 0;
  if (mx > 0.f) frexpf(mx, &e);                                 // mx = f * 2^e, 0.5 <= f < 1
  scale[0] = mx > 0.f ? ldexpf(1.f, 61 - log2_terms - e) : 1.f;
}
__global__ void ray_det_finish_kernel(const unsigned long long* detmap, const float* scale, float* dmap, int64_t dmap_sB,
                                      int64_t per_b, int B) {
  const double inv = 1.0 / (double)scale[0];
  for (int b = 0; b < B; ++b)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_b; i += (int64_t)gridDim.x * blockDim.x)
      dmap[b * dmap_sB + i] += (float)((double)(long long)detmap[b * per_b + i] * inv);
}
void* g_ray_det_buf = nullptr;
size_t g_ray_det_bytes = 0;

void* g_ray_idx_buf = nullptr;         // index scratch of crn_ray_sample_bwd (grow-only: a captured graph may hold the pointer)
size_t g_ray_idx_bytes = 0;

int ray_fwd(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP, int B, int C, int h, int w, const float* matrix,
            const float* offset, float* out, int64_t out_sB, int D, int H, int W, uint16_t* idx, hipStream_t st) {
  if (!map || !out || B < 1 || C < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1 || map_sC < 1 || map_sP < 1)
    return CRN_EINVAL;
  if (idx && (int64_t)h * w >= 65535) return CRN_EINVAL;
  const bool v4 = (W % 4 == 0) && (out_sB % 4 == 0) && (((uintptr_t)out & 15) == 0) && (((uintptr_t)idx & 7) == 0);
  // channel-last map with 16-byte aligned pixels: dwordx4 gathers
  const bool cl = map_sC == 1 && (map_sP % 4 == 0) && (map_sB % 4 == 0) && (C % 4 == 0) && (((uintptr_t)map & 15) == 0);
  const int64_t per_b = (int64_t)D * H * (v4 ? W / 4 : W);
  dim3 grid((unsigned)crn_cdiv(per_b, 256), (unsigned)B, (unsigned)std::max(1, C / 12));
#define CRN_RAY_LAUNCH(VX, CL)                                                                                      \
  hipLaunchKernelGGL((ray_sample_fwd_kernel<VX, CL>), grid, dim3(256), 0, st, map, map_sB, map_sC, map_sP, C, h, w, \
                     matrix, offset, out, out_sB, D, H, W, per_b, idx)
  if (v4 && cl) CRN_RAY_LAUNCH(4, true);
  else if (v4) CRN_RAY_LAUNCH(4, false);
  else if (cl) CRN_RAY_LAUNCH(1, true);
  else CRN_RAY_LAUNCH(1, false);
#undef CRN_RAY_LAUNCH
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

template <typename IT>
int ray_project(const float* matrix, const float* offset, int B, int D, int H, int W, int h, int w, IT* idx, hipStream_t st) {
  const bool v4 = W % 4 == 0;
  const int64_t per_b = (int64_t)D * H * (v4 ? W / 4 : W);
  dim3 grid((unsigned)crn_cdiv(per_b, 256), (unsigned)B);
  if (v4) hipLaunchKernelGGL((ray_project_kernel<4, IT>), grid, dim3(256), 0, st, matrix, offset, idx, D, H, W, h, w, per_b);
  else hipLaunchKernelGGL((ray_project_kernel<1, IT>), grid, dim3(256), 0, st, matrix, offset, idx, D, H, W, h, w, per_b);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

template <typename IT>
int ray_scatter(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W, const IT* idx, float* dmap,
                int64_t dmap_sB, int h, int w, int zero_first, hipStream_t st) {
  // A map of width 1 (the stage-5 map of a 64 x 32 image): ceil(2^32 / 1) does not fit wrecip's 32 bits.  The kernel only
  // ever splits the FLAT pixel index po = iy * w + ix of a [C][h][w] map, so a [h][1] map is scattered as the [1][h] map it
  // is in memory (iy = 0, ix = po): same addresses, and every reciprocal fits (h = w = 1 has po = 0 only).
  if (w == 1 && h > 1) { w = h; h = 1; }
  if (zero_first) {
    if (dmap_sB == (int64_t)C * h * w) {
      CRN_HIP(hipMemsetAsync(dmap, 0, (size_t)B * C * h * w * 4, st));
    } else {
      for (int b = 0; b < B; ++b) CRN_HIP(hipMemsetAsync(dmap + b * dmap_sB, 0, (size_t)C * h * w * 4, st));
    }
  }
  // tuning aids (tools/ray_bwd_sweep.sh): CRN_RAY_ZSEG = planes per thread (8 or 16), CRN_RAY_CN = channels per thread (4 or 12),
  // CRN_RAY_TX = tile width for grids >= 64 (32 or 64)
  static const int zseg_env = getenv("CRN_RAY_ZSEG") ? atoi(getenv("CRN_RAY_ZSEG")) : 0;
  static const int cn_env = getenv("CRN_RAY_CN") ? atoi(getenv("CRN_RAY_CN")) : 0;
  static const int tx_env = getenv("CRN_RAY_TX") ? atoi(getenv("CRN_RAY_TX")) : 0;
  const bool det = crn_deterministic();
  const int CN = (C % 12 == 0 && (cn_env == 12 || (det && !cn_env))) ? 12 : 4;
  // measured (profiles/r05_ray_sweep.txt, B = 4, us at 64^3 / 32^3): 4 channels x 8 planes 17.9 / 14.4, x 16 planes 26.5 / 16.1,
  // 12 channels x 8 planes 22.9 / 20.8; 64-wide tiles 17.4 / 14.4
  const int ZS = CN == 4 && zseg_env == 16 ? 16 : 8;      // (12 channels x 16 planes: 255 registers)
  const int zseg = std::min(D, ZS);
  const int nseg = (D + zseg - 1) / zseg;
  const bool big = W >= 64 && H >= 64;
  const int kTX = big ? (tx_env == 64 ? 64 : 32) : 8, kTY = big ? 256 / kTX : 8;
  const int tilesX = crn_cdiv(W, kTX), tilesY = crn_cdiv(H, kTY);
  // LDS window: the pixels a tile can reach when a voxel covers s pixels (s = map size / grid size, >= 1), with margins; a
  // tile that reaches further adds to HBM directly
  const double sx = std::max(1.0, (double)w / W), sy = std::max(1.0, (double)h / H);
  int64_t wpix = (int64_t)(kTX * sx + 4) * (int64_t)(kTY * sy + 4);
  int win_floats = det ? 0 : (int)std::min<int64_t>(wpix * CN, 6 * 1024);     // entries of 8 bytes: <= 48 KiB
  ScatterArgs a{dout, dout_sB, C, D, H, W, idx, dmap, dmap_sB, h, w, (unsigned)((((uint64_t)1 << 32) + w - 1) / w),
                zseg, tilesX, tilesY, kTX, kTY, win_floats, nullptr, nullptr, 0};
#ifdef CRN_TOOLS
  a.dbg = getenv("CRN_RAY_DBG") ? atoi(getenv("CRN_RAY_DBG")) : 0;
#endif
  dim3 grid((unsigned)(tilesX * tilesY * nseg), (unsigned)crn_cdiv(C, CN), (unsigned)B), block((unsigned)(kTX * kTY));
  const size_t lds = (size_t)win_floats * 8;
  if (det) {
    // 64-bit fixed-point accumulation: scratch = int64 image + max bits + scale
    const int64_t per_b = (int64_t)C * h * w;
    const size_t need = (size_t)B * per_b * 8 + 256;
    if (need > g_ray_det_bytes) {
      // grow-only, the outgrown buffer is kept (a captured graph may hold its address)
      CRN_HIP(hipMalloc(&g_ray_det_buf, need));
      g_ray_det_bytes = need;
    }
    unsigned long long* detmap = reinterpret_cast<unsigned long long*>(g_ray_det_buf);
    unsigned* maxbits = reinterpret_cast<unsigned*>(detmap + (size_t)B * per_b);
    float* scale = reinterpret_cast<float*>(maxbits + 16);
    CRN_HIP(hipMemsetAsync(g_ray_det_buf, 0, need, st));
    const int64_t dper_b = (int64_t)C * D * H * W;
    hipLaunchKernelGGL(ray_det_maxabs_kernel, dim3(1024), dim3(256), 0, st, dout, dout_sB, dper_b, B, maxbits);
    int lg = 0;
    while (((int64_t)1 << lg) < (int64_t)D * H * W) ++lg;
    hipLaunchKernelGGL(ray_det_scale_kernel, dim3(1), dim3(1), 0, st, maxbits, scale, lg);
    a.detmap = detmap; a.scale_p = scale;
    if (CN == 12) hipLaunchKernelGGL((ray_scatter_kernel<12, 8, IT, true>), grid, block, 0, st, a);
    else if (ZS == 8) hipLaunchKernelGGL((ray_scatter_kernel<4, 8, IT, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((ray_scatter_kernel<4, 16, IT, true>), grid, block, 0, st, a);
    hipLaunchKernelGGL(ray_det_finish_kernel, dim3(256), dim3(256), 0, st, detmap, scale, dmap, dmap_sB, per_b, B);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  if (CN == 12) hipLaunchKernelGGL((ray_scatter_kernel<12, 8, IT, false>), grid, block, lds, st, a);
  else if (ZS == 8) hipLaunchKernelGGL((ray_scatter_kernel<4, 8, IT, false>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((ray_scatter_kernel<4, 16, IT, false>), grid, block, lds, st, a);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

}  // namespace

extern "C" int crn_ray_sample_fwd(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP, int B, int C,
                                  int h, int w, const float* matrix, const float* offset, float* out,
                                  int64_t out_sB, int D, int H, int W, crnStream stream) {
  CRN_ENTRY(stream);
  return ray_fwd(map, map_sB, map_sC, map_sP, B, C, h, w, matrix, offset, out, out_sB, D, H, W, nullptr, (hipStream_t)stream);
}

extern "C" int crn_ray_sample_fwd_idx(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP, int B, int C,
                                      int h, int w, const float* matrix, const float* offset, float* out,
                                      int64_t out_sB, int D, int H, int W, uint16_t* idx, crnStream stream) {
  CRN_ENTRY(stream);
  if (!idx) return CRN_EINVAL;
  return ray_fwd(map, map_sB, map_sC, map_sP, B, C, h, w, matrix, offset, out, out_sB, D, H, W, idx, (hipStream_t)stream);
}

extern "C" int crn_ray_project(const float* matrix, const float* offset, int B, int D, int H, int W, int h, int w,
                               uint16_t* idx, crnStream stream) {
  CRN_ENTRY(stream);
  if (!matrix || !offset || !idx || B < 1 || D < 1 || H < 1 || W < 1 || h < 1 || w < 1 || (int64_t)h * w >= 65535)
    return CRN_EINVAL;
  return ray_project<uint16_t>(matrix, offset, B, D, H, W, h, w, idx, (hipStream_t)stream);
}

extern "C" int crn_ray_sample_bwd_idx(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                                      const uint16_t* idx, float* dmap, int64_t dmap_sB, int h, int w, int zero_first,
                                      crnStream stream) {
  CRN_ENTRY(stream);
  if (!dout || !dmap || !idx || B < 1 || C < 1 || D < 1 || H < 1 || W < 1 || h < 1 || w < 1 || (int64_t)h * w >= 65535)
    return CRN_EINVAL;
  return ray_scatter<uint16_t>(dout, dout_sB, B, C, D, H, W, idx, dmap, dmap_sB, h, w, zero_first, (hipStream_t)stream);
}

extern "C" int crn_ray_sample_bwd(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                                  const float* matrix, const float* offset, float* dmap,
                                  int64_t dmap_sB, int h, int w, int zero_first, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!dout || !dmap || !matrix || !offset || B < 1 || C < 1 || D < 1 || H < 1 || W < 1 || h < 1 || w < 1) return CRN_EINVAL;
  // no saved index tensor: project into the library's scratch, then the same scatter
  const bool small = (int64_t)h * w < 65535;
  const size_t need = (size_t)B * D * H * W * (small ? 2 : 4);
  if (need > g_ray_idx_bytes) {
    CRN_HIP(hipMalloc(&g_ray_idx_buf, need));      // grow-only; the outgrown buffer is kept (see g_ray_idx_buf)
    g_ray_idx_bytes = need;
  }
  if (small) {
    const int rc = ray_project<uint16_t>(matrix, offset, B, D, H, W, h, w, (uint16_t*)g_ray_idx_buf, st);
    if (rc != CRN_OK) return rc;
    return ray_scatter<uint16_t>(dout, dout_sB, B, C, D, H, W, (const uint16_t*)g_ray_idx_buf, dmap, dmap_sB, h, w, zero_first, st);
  }
  const int rc = ray_project<int32_t>(matrix, offset, B, D, H, W, h, w, (int32_t*)g_ray_idx_buf, st);
  if (rc != CRN_OK) return rc;
  return ray_scatter<int32_t>(dout, dout_sB, B, C, D, H, W, (const int32_t*)g_ray_idx_buf, dmap, dmap_sB, h, w, zero_first, st);
}
