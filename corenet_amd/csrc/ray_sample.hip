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
    int64_t per_b) {
  crn_kernargs_now(map, map_sB, map_sC, map_sP, C, h, w, matrix, offset, out, out_sB, D, H, W, per_b);
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

// un-truncated pixel coordinates (fu, fv) and the w clip coordinate: used only to bound the pixel
// window of a voxel box (a projective map of a box has its extremes at the corners when pw keeps
// its sign); the sampling decision itself is always project().
__device__ __forceinline__ void project_uv(const Cam& c, float x, float y, float z, int w, int h, float& fu,
                                           float& fv, float& pw) {
  const float cx = x + c.ox, cy = y + c.oy, cz = z + c.oz;
  const float px = row_dot(c.m + 0, cx, cy, cz), py = row_dot(c.m + 4, cx, cy, cz);
  pw = row_dot(c.m + 12, cx, cy, cz);
  fu = (px / pw * 0.5f + 0.5f) * (float)w;
  fv = (py / pw * 0.5f + 0.5f) * (float)h;
}

// backward: a workgroup owns a TX x TY tile of (x,y) voxel columns over a z segment; one thread =
// one column and CN channels (12 = every skip width, so the projection is evaluated once per voxel).
// Consecutive z usually hit the same pixel -> run-length accumulate in registers.  Neighbouring
// columns hit the SAME pixels, and same-address float atomics in L2 serialise (measured: 100 us
// with them, 13 us without): runs are therefore added into an LDS copy of the pixel window that
// the tile can reach (bounding box of the projected box corners), and the window goes to HBM
// with one atomic per touched (pixel, channel).  Pixels outside the window (box straddling the
// camera plane, or a window larger than the LDS budget) fall back to global atomics.
// Tried and dropped in round 2: a gather by PIXEL OWNERS (one thread per pixel x depth x 12 channels inverts the
// projection of a view-space camera -- clip x depends on (x, z) only, clip y on (y, z), clip w on z -- finds the
// few voxels that land in its pixel with the same bit-defined project(), sums them, no atomics, no memset).  Bit
// for bit correct, but 56 / 19 / 18 / 18 us at the four scales against 49 / 24 / 16 / 14 us here: the exact
// membership test needs ~8 IEEE divisions per (pixel, z), the reads come in 64-byte pieces, and general cameras
// still need this kernel as a second launch.
// Tried and dropped in round 4: a LOAD-FIRST form (a thread owns 4 consecutive x columns x 4 channels x 8 planes and issues all
// of its 32 16-byte loads before it touches camera, window box or LDS; the eight z segments of a tile are the eight waves
// of ONE workgroup and share one 16 KiB window, which goes to HBM once per tile: 8x fewer device-scope atomics; the box from
// 8 lanes at once).  Correct on every test, and slower: 50 / 26 / 21 / 18 us (burst incl. the 4 us memset) against
// 44 / 17 / 14 / 14 us here.  Its ablations at 64^3 say where the time of BOTH kernels is: 29 us with the projection
// replaced by a constant pixel (one flush per column: loads + the shell of the launch), 30 us with the projection but
// without the LDS adds, 51 us with them -- the ~36 divergent flush sites of a wave (4 columns x 9) each issue 4
// ds_add_f32 for whatever lanes changed pixel at that plane, ~1150 LDS atomic instructions per CU; the window write-out
// (3 us) and the tile shape (whole 256-byte rows: the same 28 us floor) do not matter.  Fewer, fuller LDS atomic
// INSTRUCTIONS do not help either: with the last two runs of a column kept as records in registers and flushed together at
// the end of the segment (two nearly full flush sites instead of nine sparse ones) this kernel takes 48 us instead of 44 --
// the cost follows the lane-adds (4.7 M at 64^3, same-pixel neighbours serialise), not the instruction count.
constexpr int kTMax = 16;                      // tile: 32x8 columns (full 128-B rows) for grids >= 64, else 8x8
constexpr int kWinFloats = 12 * 1024;          // 48 KiB of LDS
// DET (deterministic mode, crn_common.h): the sums are taken in 64-bit fixed point -- integer adds commute, so the
// result does not depend on the order in which threads reach a pixel.  scale_p[0] = 2^k chosen from max |dout| so
// that a pixel's sum cannot overflow; detmap = int64 image of dmap ([B][C][h][w], zeroed), converted by
// ray_det_finish_kernel.  The LDS window is not used (its float adds are the order-dependent part).
template <int CN, bool DET>
__global__ __launch_bounds__(kTMax * kTMax) void ray_sample_bwd_kernel(
    const float* __restrict__ dout, int64_t dout_sB, int C, int D, int H, int W, const float* matrix,
    const float* offset, float* dmap, int64_t dmap_sB, int h, int w, int zseg, int tilesX, int tilesY, int kTX, int kTY,
    unsigned long long* detmap, const float* scale_p) {
  __shared__ float win[kWinFloats];
  __shared__ int wbox[4];
  crn_kernargs_now(dout, dout_sB, C, D, H, W, matrix, offset, dmap, dmap_sB, h, w, zseg, tilesX, tilesY, kTX, kTY);
  const int b = blockIdx.z;
  const int cbase = blockIdx.y * CN;
  int tile = blockIdx.x;
  const int tx = tile % tilesX; tile /= tilesX;
  const int ty = tile % tilesY; tile /= tilesY;
  const int z0 = tile * zseg, z1 = min(D, z0 + zseg);
  const int x0 = tx * kTX, y0 = ty * kTY;
  const int x = x0 + (int)(threadIdx.x % kTX), y = y0 + (int)(threadIdx.x / kTX);
  const Cam cam = load_cam(matrix, offset, b);
  if (threadIdx.x == 0) {
    const int x1 = min(W, x0 + kTX) - 1, y1 = min(H, y0 + kTY) - 1;
    float umin = 3e38f, umax = -3e38f, vmin = 3e38f, vmax = -3e38f;
    bool front = true;
    for (int k = 0; k < 8; ++k) {
      float fu, fv, pw;
      project_uv(cam, (float)((k & 1) ? x1 : x0), (float)((k & 2) ? y1 : y0), (float)((k & 4) ? z1 - 1 : z0), w, h,
                 fu, fv, pw);
      front = front && pw > 1e-6f;
      umin = fminf(umin, fu); umax = fmaxf(umax, fu); vmin = fminf(vmin, fv); vmax = fmaxf(vmax, fv);
    }
    int ix0 = 0, ix1 = -1, iy0 = 0, iy1 = -1;                 // empty window
    if (front && umax > -1e6f && umin < 1e6f && vmax > -1e6f && vmin < 1e6f) {
      ix0 = max(0, (int)floorf(umin) - 1); ix1 = min(w - 1, (int)ceilf(umax) + 1);
      iy0 = max(0, (int)floorf(vmin) - 1); iy1 = min(h - 1, (int)ceilf(vmax) + 1);
      if (ix1 < ix0 || iy1 < iy0 || (int64_t)(ix1 - ix0 + 1) * (iy1 - iy0 + 1) * CN > kWinFloats) { ix1 = -1; iy1 = -1; ix0 = iy0 = 0; }
    }
    if (DET) { ix0 = iy0 = 0; ix1 = iy1 = -1; }              // no LDS window: every run goes to the fixed-point image
    wbox[0] = ix0; wbox[1] = ix1; wbox[2] = iy0; wbox[3] = iy1;
  }
  __syncthreads();
  const int ix0 = wbox[0], ix1 = wbox[1], iy0 = wbox[2], iy1 = wbox[3];
  const int ww = ix1 - ix0 + 1, wh = iy1 - iy0 + 1, wn = ww > 0 && wh > 0 ? ww * wh : 0;
  for (int i = threadIdx.x; i < wn * CN; i += blockDim.x) win[i] = 0.f;
  __syncthreads();
  const int64_t S = (int64_t)D * H * W, hw = (int64_t)h * w;
  float* mb = dmap + (int64_t)b * dmap_sB + (int64_t)cbase * hw;
  if (x < W && y < H) {
    const float* gb = dout + (int64_t)b * dout_sB + (int64_t)cbase * S + (int64_t)y * W + x;
    float acc[CN];
#pragma unroll
    for (int k = 0; k < CN; ++k) acc[k] = 0.f;
    int cur = -1;
    auto flush = [&]() {
      if (cur >= 0) {
        const int iy = cur / w, ix = cur - iy * w;
        if (ix >= ix0 && ix <= ix1 && iy >= iy0 && iy <= iy1) {
          float* wp = win + (iy - iy0) * ww + (ix - ix0);
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(wp + k * wn, acc[k]);          // ds_add_f32
        } else if (DET) {
          const float sc = scale_p[0];
          unsigned long long* ib = detmap + ((int64_t)b * C + cbase) * hw;
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(ib + k * hw + cur, (unsigned long long)(long long)__float2ll_rn(acc[k] * sc));
        } else {
#pragma unroll
          for (int k = 0; k < CN; ++k)
            if (cbase + k < C) atomicAdd(mb + k * hw + cur, acc[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < CN; ++k) acc[k] = 0.f;
    };
    // The z loop is a chain of (load, project, compare, maybe flush) steps whose branches keep the compiler from
    // hoisting the next loads: with one wave per SIMD (256 workgroups at 64^3) every step paid a full HBM latency
    // (49 us for 50 MB).  Loads are therefore issued kZC planes at a time into two register buffers, the next
    // chunk in flight while the current one is consumed; addresses are clamped so that no load sits under a branch.
    constexpr int kZC = 4;
    float ga[kZC][CN], gb2[kZC][CN];
    auto load_chunk = [&](float (&g)[kZC][CN], int zc) {
#pragma unroll
      for (int j = 0; j < kZC; ++j) {
        const int64_t zo = (int64_t)min(zc + j, z1 - 1) * H * W;
#pragma unroll
        for (int k = 0; k < CN; ++k) g[j][k] = __builtin_nontemporal_load(gb + (cbase + k < C ? k : 0) * S + zo);
      }
    };
    auto consume = [&](float (&g)[kZC][CN], int zc) {
#pragma unroll
      for (int j = 0; j < kZC; ++j) {
        const int z = zc + j;
        if (z < z1) {
          const int po = project(cam, x, y, z, w, h);
          if (po != cur) { flush(); cur = po; }
          if (po >= 0) {
#pragma unroll
            for (int k = 0; k < CN; ++k) acc[k] += g[j][k];
          }
        }
      }
    };
    load_chunk(ga, z0);
    for (int zc = z0; zc < z1; zc += 2 * kZC) {
      load_chunk(gb2, zc + kZC);
      consume(ga, zc);
      load_chunk(ga, zc + 2 * kZC);
      consume(gb2, zc + kZC);
    }
    flush();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < wn * CN; i += blockDim.x) {
    const float v = win[i];
    if (v != 0.f) {
      const int k = i / wn, p = i - k * wn;
      const int iy = iy0 + p / ww, ix = ix0 + p % ww;
      if (cbase + k < C) atomicAdd(mb + k * hw + (int64_t)iy * w + ix, v);
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

}  // namespace

extern "C" int crn_ray_sample_fwd(const float* map, int64_t map_sB, int64_t map_sC, int64_t map_sP, int B, int C,
                                  int h, int w, const float* matrix, const float* offset, float* out,
                                  int64_t out_sB, int D, int H, int W, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!map || !out || B < 1 || C < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1 || map_sC < 1 || map_sP < 1)
    return CRN_EINVAL;
  const bool v4 = (W % 4 == 0) && (out_sB % 4 == 0) && (((uintptr_t)out & 15) == 0);
  // channel-last map with 16-byte aligned pixels: dwordx4 gathers
  const bool cl = map_sC == 1 && (map_sP % 4 == 0) && (map_sB % 4 == 0) && (C % 4 == 0) && (((uintptr_t)map & 15) == 0);
  const int64_t per_b = (int64_t)D * H * (v4 ? W / 4 : W);
  dim3 grid((unsigned)crn_cdiv(per_b, 256), (unsigned)B, (unsigned)std::max(1, C / 12));
#define CRN_RAY_LAUNCH(VX, CL)                                                                                      \
  hipLaunchKernelGGL((ray_sample_fwd_kernel<VX, CL>), grid, dim3(256), 0, st, map, map_sB, map_sC, map_sP, C, h, w, \
                     matrix, offset, out, out_sB, D, H, W, per_b)
  if (v4 && cl) CRN_RAY_LAUNCH(4, true);
  else if (v4) CRN_RAY_LAUNCH(4, false);
  else if (cl) CRN_RAY_LAUNCH(1, true);
  else CRN_RAY_LAUNCH(1, false);
#undef CRN_RAY_LAUNCH
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_ray_sample_bwd(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                                  const float* matrix, const float* offset, float* dmap,
                                  int64_t dmap_sB, int h, int w, int zero_first, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!dout || !dmap || B < 1 || C < 1) return CRN_EINVAL;
  if (zero_first) {
    if (dmap_sB == (int64_t)C * h * w) {
      CRN_HIP(hipMemsetAsync(dmap, 0, (size_t)B * C * h * w * 4, st));
    } else {
      for (int b = 0; b < B; ++b) CRN_HIP(hipMemsetAsync(dmap + b * dmap_sB, 0, (size_t)C * h * w * 4, st));
    }
  }
  static const int zseg_env = getenv("CRN_RAY_ZSEG") ? atoi(getenv("CRN_RAY_ZSEG")) : 0;
  // z segment and channels per thread, measured at the four decoder scales (tools/ray_bwd_sweep.sh,
  // profiles/r03_ray_bwd_sweep.txt): 4 channels per thread (three times the waves of the 12-channel variant: the kernel is
  // latency bound, one workgroup of four waves per CU at 64^3) and z segments of 8 (16 at 32^3):
  // 64^3 48.7 -> 44.4 us, 32^3 24.0 -> 17.3 us, 16^3 19.0 -> 11.7 us, 8^3 13.2 -> 11.5 us
  const int zseg = zseg_env > 0 ? std::min(zseg_env, D) : (D == 32 ? 16 : std::min(D, 8));
  const int nseg = (D + zseg - 1) / zseg;
  const bool big = W >= 64 && H >= 64;
  const int kTX = big ? 32 : 8, kTY = 8;
  const int tilesX = crn_cdiv(W, kTX), tilesY = crn_cdiv(H, kTY);
  if (crn_deterministic()) {
    // 64-bit fixed-point accumulation (see ray_sample_bwd_kernel<.., DET>): scratch = int64 image + max bits + scale
    const int64_t per_b = (int64_t)C * h * w;
    const size_t need = (size_t)B * per_b * 8 + 256;
    if (need > g_ray_det_bytes) {
      if (g_ray_det_buf) CRN_HIP(hipFree(g_ray_det_buf));
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
    if (C % 12 == 0) {
      dim3 grid((unsigned)(tilesX * tilesY * nseg), (unsigned)(C / 12), (unsigned)B);
      hipLaunchKernelGGL((ray_sample_bwd_kernel<12, true>), grid, dim3(kTX * kTY), 0, st, dout, dout_sB, C, D, H, W, matrix,
                         offset, dmap, dmap_sB, h, w, zseg, tilesX, tilesY, kTX, kTY, detmap, scale);
    } else {
      dim3 grid((unsigned)(tilesX * tilesY * nseg), (unsigned)crn_cdiv(C, 4), (unsigned)B);
      hipLaunchKernelGGL((ray_sample_bwd_kernel<4, true>), grid, dim3(kTX * kTY), 0, st, dout, dout_sB, C, D, H, W, matrix,
                         offset, dmap, dmap_sB, h, w, zseg, tilesX, tilesY, kTX, kTY, detmap, scale);
    }
    hipLaunchKernelGGL(ray_det_finish_kernel, dim3(256), dim3(256), 0, st, detmap, scale, dmap, dmap_sB, per_b, B);
    CRN_CHECK_LAUNCH();
    return CRN_OK;
  }
  static const bool cn4 = !(getenv("CRN_RAY_CN") && atoi(getenv("CRN_RAY_CN")) == 12);
  if (C % 12 == 0 && !cn4) {
    dim3 grid((unsigned)(tilesX * tilesY * nseg), (unsigned)(C / 12), (unsigned)B);
    hipLaunchKernelGGL((ray_sample_bwd_kernel<12, false>), grid, dim3(kTX * kTY), 0, st, dout, dout_sB, C, D, H, W, matrix,
                       offset, dmap, dmap_sB, h, w, zseg, tilesX, tilesY, kTX, kTY, (unsigned long long*)nullptr, (const float*)nullptr);
  } else {
    dim3 grid((unsigned)(tilesX * tilesY * nseg), (unsigned)crn_cdiv(C, 4), (unsigned)B);
    hipLaunchKernelGGL((ray_sample_bwd_kernel<4, false>), grid, dim3(kTX * kTY), 0, st, dout, dout_sB, C, D, H, W, matrix,
                       offset, dmap, dmap_sB, h, w, zseg, tilesX, tilesY, kTX, kTY, (unsigned long long*)nullptr, (const float*)nullptr);
  }
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
