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
  const float u = __fadd_rn(__fdiv_rn(__fdiv_rn(px, pw), 2.0f), 0.5f);
  const float v = __fadd_rn(__fdiv_rn(__fdiv_rn(py, pw), 2.0f), 0.5f);
  const float fu = __fmul_rn(u, (float)w), fv = __fmul_rn(v, (float)h);
  // (int) cast: truncation; out-of-range values saturate and fall outside [0,w)
  const int ix = (int)fu, iy = (int)fv;
  const bool ok = (pz >= 0.0f) && ix >= 0 && ix < w && iy >= 0 && iy < h &&
                  fu < 2147483520.0f && fv < 2147483520.0f && fu > -2147483520.0f && fv > -2147483520.0f;
  return ok ? iy * w + ix : -1;
}

// forward: one thread = 4 consecutive x voxels, loops over channels
template <int VX>
__global__ __launch_bounds__(256) void ray_sample_fwd_kernel(
    const float* __restrict__ map, int64_t map_sB, int C, int h, int w, const float* matrix,
    const float* offset, float* __restrict__ out, int64_t out_sB, int D, int H, int W, int64_t per_b) {
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
  const int64_t hw = (int64_t)h * w;
  const float* mb = map + (int64_t)b * map_sB;
  float* ob = out + (int64_t)b * out_sB + ((int64_t)z * H + y) * W + x0;
  // 12 channels per iteration (every skip width 96/48/24/12 is a multiple): 48 independent gathers in
  // flight per lane, then 12 streaming (non-temporal) float4 stores -- the output is consumed much
  // later by the next decoder stage, so it should not displace the feature map from L2.
  int c = 0;
  for (; c + 12 <= C; c += 12) {
    float v[12][VX];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const float* mc = mb + (c + u) * hw;
#pragma unroll
      for (int k = 0; k < VX; ++k) v[u][k] = po[k] >= 0 ? mc[po[k]] : 0.f;
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
  for (; c < C; ++c) {
    const float* mc = mb + c * hw;
    float v[VX];
#pragma unroll
    for (int k = 0; k < VX; ++k) v[k] = po[k] >= 0 ? mc[po[k]] : 0.f;
    if (VX == 4) {
      *reinterpret_cast<f32x4*>(ob + c * S) = (f32x4){v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
      for (int k = 0; k < VX; ++k) ob[c * S + k] = v[k];
    }
  }
}

// backward: one thread = one (x,y) voxel column over a z segment and CN channels (12 = every skip
// width, so the projection is evaluated once per voxel like in the forward); consecutive z usually
// hit the same pixel -> run-length accumulate in registers, one atomic per run and channel.
template <int CN>
__global__ __launch_bounds__(256) void ray_sample_bwd_kernel(
    const float* __restrict__ dout, int64_t dout_sB, int C, int D, int H, int W, const float* matrix,
    const float* offset, float* dmap, int64_t dmap_sB, int h, int w, int zseg, int dbg) {
  const int b = blockIdx.z;
  const int cbase = blockIdx.y * CN;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t cols = (int64_t)H * W;
  const int nseg = (D + zseg - 1) / zseg;
  if (t >= cols * nseg) return;
  const int x = (int)(t % W);
  const int y = (int)((t / W) % H);
  const int z0 = (int)(t / cols) * zseg;
  const int z1 = min(D, z0 + zseg);
  const Cam cam = load_cam(matrix, offset, b);
  const int64_t S = (int64_t)D * H * W, hw = (int64_t)h * w;
  const float* gb = dout + (int64_t)b * dout_sB + (int64_t)cbase * S + (int64_t)y * W + x;
  float* mb = dmap + (int64_t)b * dmap_sB + (int64_t)cbase * hw;
  float acc[CN];
#pragma unroll
  for (int k = 0; k < CN; ++k) acc[k] = 0.f;
  int cur = -1;
  auto flush = [&]() {
    if (cur >= 0) {
#pragma unroll
      for (int k = 0; k < CN; ++k)
        if (cbase + k < C && dbg != 1) atomicAdd(mb + k * hw + cur, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < CN; ++k) acc[k] = 0.f;
  };
  for (int z = z0; z < z1; ++z) {
    // the CN loads of this step do not depend on the projection: issue them first
    float gv[CN];
#pragma unroll
    for (int k = 0; k < CN; ++k)
      gv[k] = (cbase + k < C && dbg != 2) ? __builtin_nontemporal_load(gb + k * S + (int64_t)z * H * W) : 1.f;
    const int po = dbg == 3 ? (y * w + x) : project(cam, x, y, z, w, h);
    if (po != cur) { flush(); cur = po; }
    if (po >= 0) {
#pragma unroll
      for (int k = 0; k < CN; ++k) acc[k] += gv[k];
    }
  }
  flush();
}

}  // namespace

extern "C" int crn_ray_sample_fwd(const float* map, int64_t map_sB, int B, int C, int h, int w,
                                  const float* matrix, const float* offset, float* out,
                                  int64_t out_sB, int D, int H, int W, crnStream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!map || !out || B < 1 || C < 1 || h < 1 || w < 1 || D < 1 || H < 1 || W < 1) return CRN_EINVAL;
  const bool v4 = (W % 4 == 0) && (out_sB % 4 == 0) && (((uintptr_t)out & 15) == 0);
  const int64_t per_b = (int64_t)D * H * (v4 ? W / 4 : W);
  dim3 grid((unsigned)crn_cdiv(per_b, 256), (unsigned)B);
  if (v4)
    hipLaunchKernelGGL(ray_sample_fwd_kernel<4>, grid, dim3(256), 0, st, map, map_sB, C, h, w, matrix, offset,
                       out, out_sB, D, H, W, per_b);
  else
    hipLaunchKernelGGL(ray_sample_fwd_kernel<1>, grid, dim3(256), 0, st, map, map_sB, C, h, w, matrix, offset,
                       out, out_sB, D, H, W, per_b);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_ray_sample_bwd(const float* dout, int64_t dout_sB, int B, int C, int D, int H, int W,
                                  const float* matrix, const float* offset, float* dmap,
                                  int64_t dmap_sB, int h, int w, int zero_first, crnStream stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!dout || !dmap || B < 1 || C < 1) return CRN_EINVAL;
  if (zero_first) {
    if (dmap_sB == (int64_t)C * h * w) {
      CRN_HIP(hipMemsetAsync(dmap, 0, (size_t)B * C * h * w * 4, st));
    } else {
      for (int b = 0; b < B; ++b) CRN_HIP(hipMemsetAsync(dmap + b * dmap_sB, 0, (size_t)C * h * w * 4, st));
    }
  }
  const int dbg = getenv("CRN_RAY_DBG") ? atoi(getenv("CRN_RAY_DBG")) : 0;
  const int zseg = getenv("CRN_RAY_ZSEG") ? atoi(getenv("CRN_RAY_ZSEG")) : (D >= 16 ? 8 : D);
  const int nseg = (D + zseg - 1) / zseg;
  if (C % 12 == 0) {
    dim3 grid((unsigned)crn_cdiv((int64_t)H * W * nseg, 256), (unsigned)(C / 12), (unsigned)B);
    hipLaunchKernelGGL(ray_sample_bwd_kernel<12>, grid, dim3(256), 0, st, dout, dout_sB, C, D, H, W, matrix, offset,
                       dmap, dmap_sB, h, w, zseg, dbg);
  } else {
    dim3 grid((unsigned)crn_cdiv((int64_t)H * W * nseg, 256), (unsigned)crn_cdiv(C, 4), (unsigned)B);
    hipLaunchKernelGGL(ray_sample_bwd_kernel<4>, grid, dim3(256), 0, st, dout, dout_sB, C, D, H, W, matrix, offset,
                       dmap, dmap_sB, h, w, zseg, dbg);
  }
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
