// Triangle-mesh surface voxelizer + per-scene label merge (ground-truth side).
// Reference: geometry/voxelization.py:98-164 driving the OpenGL shaders
// geometry/shaders/voxelize.geom:31-60 and voxelize.frag:29-58 through an
// NVIDIA-only EGL context (gl/rasterizer.py) with a host round trip per batch;
// data/batched_example.py:186-196 for the label merge.
// MI355X compute nodes have no GL: this is a software rasterizer with the GL
// rules written out (pixel-centre sampling, top-left fill rule, optional
// conservative overlap test, depth clip), one wavefront per triangle, idempotent
// stores (=1) so no atomics.  Same formulae as oracle.corenet_oracle.voxelize_mesh
// in fp32.  Parity with the hardware rasterizer is pinned only by the reference's
// three known-answer tests (voxelization_test.py:53-147).
#include "crn_common.h"
#include <algorithm>
#include <cmath>

namespace {

struct VoxParams {
  const float* tri; const int32_t* tri_mesh; int T;
  const float* v2v; int M, D, H, W;
  int vs;            // virtual_voxel_side (sub-grid) or -1
  int R;             // render target resolution
  float depth_ext;   // D * projection_depth_multiplier
  int conservative;
  float* grid;
};

__global__ __launch_bounds__(256) void voxelize_kernel(VoxParams p) {
  const int lane = threadIdx.x & 63;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= p.T) return;
  const int m = p.tri_mesh[t];
  const float* A = p.v2v + (int64_t)m * 16;
  float v[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float x = p.tri[(int64_t)t * 9 + i * 3 + 0], y = p.tri[(int64_t)t * 9 + i * 3 + 1], z = p.tri[(int64_t)t * 9 + i * 3 + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) v[i][r] = A[r * 4 + 0] * x + A[r * 4 + 1] * y + A[r * 4 + 2] * z + A[r * 4 + 3];
  }
  // voxelize.geom:47: normal of the voxel-space triangle picks the projection axis
  float e1[3], e2[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { e1[r] = v[1][r] - v[0][r]; e2[r] = v[2][r] - v[0][r]; }
  const float n1 = sqrtf(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
  const float n2 = sqrtf(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
  if (n1 == 0.f || n2 == 0.f) return;
#pragma unroll
  for (int r = 0; r < 3; ++r) { e1[r] /= n1; e2[r] /= n2; }
  const float ax_ = fabsf(e1[1] * e2[2] - e1[2] * e2[1]);
  const float ay_ = fabsf(e1[2] * e2[0] - e1[0] * e2[2]);
  const float az_ = fabsf(e1[0] * e2[1] - e1[1] * e2[0]);
  int a0 = 0, a1 = 1, a2 = 2;                      // screen x, screen y, depth <- voxel axes
  if (ax_ > ay_ && ax_ > az_) { a0 = 1; a1 = 2; a2 = 0; }        // gl_Position.yzxw
  else if (ay_ > ax_ && ay_ > az_) { a0 = 2; a1 = 0; a2 = 1; }   // gl_Position.zxyw
  // transformations.ortho_lh(0, W, H, 0, 0, depth_ext): x: 2x/W-1, y: 1-2y/H, z: 2z/De-1
  float ndc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ndc[i][0] = 2.f * v[i][0] / (float)p.W - 1.f;
    ndc[i][1] = 1.f - 2.f * v[i][1] / (float)p.H;
    ndc[i][2] = 2.f * v[i][2] / p.depth_ext - 1.f;
  }
  float sx[3], sy[3], sz[3];
  const float Rf = (float)p.R;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sx[i] = (ndc[i][a0] + 1.f) * 0.5f * Rf;
    sy[i] = (ndc[i][a1] + 1.f) * 0.5f * Rf;
    sz[i] = ndc[i][a2];
  }
  const float area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sx[2] - sx[0]) * (sy[1] - sy[0]);
  if (area == 0.f || !(area == area)) return;
  const float sgn = area > 0.f ? 1.f : -1.f;
  const int x0 = max((int)floorf(fminf(sx[0], fminf(sx[1], sx[2]))) - 1, 0);
  const int x1 = min((int)ceilf(fmaxf(sx[0], fmaxf(sx[1], sx[2]))) + 1, p.R);
  const int y0 = max((int)floorf(fminf(sy[0], fminf(sy[1], sy[2]))) - 1, 0);
  const int y1 = min((int)ceilf(fmaxf(sy[0], fmaxf(sy[1], sy[2]))) + 1, p.R);
  if (x1 <= x0 || y1 <= y0) return;
  float EA[3], EB[3], EC[3];
  bool tl[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int ja = (i + 1) % 3, jb = (i + 2) % 3;
    EA[i] = (sy[ja] - sy[jb]) * sgn;
    EB[i] = (sx[jb] - sx[ja]) * sgn;
    EC[i] = (sx[ja] * sy[jb] - sx[jb] * sy[ja]) * sgn;
    tl[i] = (EA[i] > 0.f) || (EA[i] == 0.f && EB[i] > 0.f);
  }
  const float inv_tot = 1.f / fabsf(area);
  const int bw = x1 - x0;
  const int64_t npix = (int64_t)bw * (y1 - y0);
  const int Dg = p.vs > 0 ? 2 * p.D + 1 : p.D, Hg = p.vs > 0 ? 2 * p.H + 1 : p.H, Wg = p.vs > 0 ? 2 * p.W + 1 : p.W;
  float* gm = p.grid + (int64_t)m * Dg * Hg * Wg;
  for (int64_t q = lane; q < npix; q += 64) {
    const float PX = (float)(x0 + (int)(q % bw)) + 0.5f, PY = (float)(y0 + (int)(q / bw)) + 0.5f;
    float l[3];
    bool inside = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float E = EA[i] * PX + EB[i] * PY + EC[i];
      l[i] = E * inv_tot;
      if (p.conservative) inside = inside && (E + 0.5f * (fabsf(EA[i]) + fabsf(EB[i])) >= 0.f);
      else inside = inside && (E > 0.f || (E == 0.f && tl[i]));
    }
    if (!inside) continue;
    const float zn = l[0] * sz[0] + l[1] * sz[1] + l[2] * sz[2];
    if (!(zn >= -1.f && zn <= 1.f)) continue;
    const float px = l[0] * v[0][0] + l[1] * v[1][0] + l[2] * v[2][0];
    const float py = l[0] * v[0][1] + l[1] * v[1][1] + l[2] * v[2][1];
    const float pz = l[0] * v[0][2] + l[1] * v[1][2] + l[2] * v[2][2];
    // voxelize.frag:36-40
    if (px < 0.f || py < 0.f || pz < 0.f || px >= (float)p.W || py >= (float)p.H || pz >= (float)p.D) continue;
    int cx, cy, cz;
    if (p.vs <= 0) {                                   // voxelize.frag:42-47
      cx = (int)floorf(px); cy = (int)floorf(py); cz = (int)floorf(pz);
    } else {                                           // voxelize.frag:48-56
      const int vx = (int)floorf(px * (float)p.vs) + p.vs / 2;
      const int vy = (int)floorf(py * (float)p.vs) + p.vs / 2;
      const int vz = (int)floorf(pz * (float)p.vs) + p.vs / 2;
      cx = 2 * (vx / p.vs) + ((vx % p.vs) == p.vs - 1 ? 1 : 0);
      cy = 2 * (vy / p.vs) + ((vy % p.vs) == p.vs - 1 ? 1 : 0);
      cz = 2 * (vz / p.vs) + ((vz % p.vs) == p.vs - 1 ? 1 : 0);
    }
    if (cx < Wg && cy < Hg && cz < Dg) gm[((int64_t)cz * Hg + cy) * Wg + cx] = 1.0f;
  }
}

// batched_example.py:186-196 (+ voxelization.get_sub_grid_centers :167-182)
__global__ void merge_labels_kernel(const float* mg, const int32_t* scene_start, const int32_t* label, int D,
                                    int H, int W, int sub, int32_t* out, int64_t S) {
  const int b = blockIdx.y;
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= S) return;
  int64_t src = v, Sm = S;
  if (sub) {
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
    const int Wg = 2 * W + 1, Hg = 2 * H + 1, Dg = 2 * D + 1;
    src = ((int64_t)(1 + 2 * z) * Hg + (1 + 2 * y)) * Wg + (1 + 2 * x);
    Sm = (int64_t)Dg * Hg * Wg;
  }
  float best = -INFINITY;
  for (int m = scene_start[b]; m < scene_start[b + 1]; ++m) best = fmaxf(best, (float)label[m] * mg[m * Sm + src]);
  out[(int64_t)b * S + v] = scene_start[b + 1] > scene_start[b] ? (int32_t)best : 0;
}

// batched_example.batch (batched_example.py:73-81) -> transformations.transform_mesh (:139-169): every vertex of
// mesh m is multiplied by its object->view matrix as a point (w = 1) and divided by the resulting w.
__global__ void transform_meshes_kernel(const float* __restrict__ tri, const int32_t* __restrict__ tri_mesh,
                                        const float* __restrict__ mats, int64_t nvert, float* __restrict__ out) {
  const int64_t v = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (v >= nvert) return;
  const float* M = mats + (int64_t)tri_mesh[v / 3] * 16;
  const float x = tri[v * 3], y = tri[v * 3 + 1], z = tri[v * 3 + 2];
  float r[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) r[n] = fmaf(M[n * 4 + 2], z, fmaf(M[n * 4 + 1], y, fmaf(M[n * 4], x, M[n * 4 + 3])));
  out[v * 3] = r[0] / r[3]; out[v * 3 + 1] = r[1] / r[3]; out[v * 3 + 2] = r[2] / r[3];
}

}  // namespace

extern "C" int crn_transform_meshes(const float* triangles, const int32_t* tri_mesh, int T, const float* mesh_matrix,
                                    int M, float* out, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (T < 0 || M < 0 || (T > 0 && (!triangles || !tri_mesh || !mesh_matrix || !out))) return CRN_EINVAL;
  if (T == 0) return CRN_OK;
  const int64_t nvert = (int64_t)T * 3;
  hipLaunchKernelGGL(transform_meshes_kernel, dim3((unsigned)crn_cdiv(nvert, 256)), dim3(256), 0, st, triangles,
                     tri_mesh, mesh_matrix, nvert, out);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_voxelize_mesh(const float* triangles, const int32_t* tri_mesh, int T, const float* view2voxel,
                                 int M, int D, int H, int W, int sub_grid_side, float image_resolution_multiplier,
                                 int conservative, int depth_multiplier, float* grid, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!grid || M < 1 || D < 1 || H < 1 || W < 1 || T < 0) return CRN_EINVAL;
  if (sub_grid_side > 0 && (sub_grid_side % 2 == 0)) return CRN_EINVAL;   // voxelization.py:107-109
  VoxParams p;
  p.tri = triangles; p.tri_mesh = tri_mesh; p.T = T; p.v2v = view2voxel; p.M = M; p.D = D; p.H = H; p.W = W;
  p.vs = sub_grid_side > 0 ? sub_grid_side : -1;
  p.depth_ext = (float)(D * depth_multiplier);
  const int ext = std::max(W, std::max(H, D * depth_multiplier));
  p.R = (int)std::lround((double)ext * (double)image_resolution_multiplier);   // voxelization.py:146-149
  p.conservative = conservative; p.grid = grid;
  const int64_t S = sub_grid_side > 0 ? (int64_t)(2 * D + 1) * (2 * H + 1) * (2 * W + 1) : (int64_t)D * H * W;
  CRN_HIP(hipMemsetAsync(grid, 0, (size_t)M * S * sizeof(float), st));
  if (T == 0) return CRN_OK;
  hipLaunchKernelGGL(voxelize_kernel, dim3((unsigned)crn_cdiv((int64_t)T * 64, 256)), dim3(256), 0, st, p);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}

extern "C" int crn_merge_labels(const float* meshes_grid, const int32_t* scene_mesh_start, const int32_t* mesh_label,
                                int B, int D, int H, int W, int sub_grid, int32_t* out, crnStream stream) {
  CRN_ENTRY(stream);
  hipStream_t st = (hipStream_t)stream;
  if (!meshes_grid || !out || B < 1) return CRN_EINVAL;
  const int64_t S = (int64_t)D * H * W;
  hipLaunchKernelGGL(merge_labels_kernel, dim3((unsigned)crn_cdiv(S, 256), (unsigned)B), dim3(256), 0, st, meshes_grid,
                     scene_mesh_start, mesh_label, D, H, W, sub_grid, out, S);
  CRN_CHECK_LAUNCH();
  return CRN_OK;
}
