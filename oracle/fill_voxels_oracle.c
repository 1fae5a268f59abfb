/* CPU ORACLE (test infrastructure only) for fill_inside_voxels.
 *
 * Restates the algorithm of the reference's CPU op
 * (cc/fill_voxels_cpu.cc:74-142: raster-scan connected components with a
 * union-find over the -x/-y/-z neighbours, region 0 = outside, reachable only
 * through the low faces) with the OUTPUT semantics of the GPU op the pipeline
 * uses (cc/fill_voxels_gpu.cu:122-132: every voxel becomes root==0 ? 0 : 1).
 * Plain C, float grids [N][D][H][W].  Build: gcc -O2 -shared -fPIC.
 */
#include <stdint.h>
#include <stdlib.h>

static int64_t find_root(int64_t* parent, int64_t e) {
  int64_t r = e;
  while (parent[r] != -1) r = parent[r];
  while (parent[e] != -1) { int64_t n = parent[e]; parent[e] = r; e = n; }   /* path compression (:47-61) */
  return r;
}

static void merge(int64_t* parent, int64_t a, int64_t b) {   /* :36-45: smaller id becomes root */
  a = find_root(parent, a); b = find_root(parent, b);
  if (a < b) parent[b] = a; else if (b < a) parent[a] = b;
}

static void fill_one(const float* vol, float* out, int D, int H, int W) {
  const int64_t size = (int64_t)D * H * W;
  int64_t* regions = (int64_t*)malloc(sizeof(int64_t) * size);
  int64_t* parent = (int64_t*)malloc(sizeof(int64_t) * (size + 1));
  int64_t nreg = 1;
  parent[0] = -1;                                   /* region 0: outside (:80-82) */
  for (int z = 0; z < D; z++)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        const int64_t cur = ((int64_t)z * H + y) * W + x;
        const int vc = vol[cur] > 0;
        const int vl = x > 0 ? vol[cur - 1] > 0 : 0;          /* outside value is "empty" (:94-99) */
        const int vb = y > 0 ? vol[cur - W] > 0 : 0;
        const int vu = z > 0 ? vol[cur - (int64_t)W * H] > 0 : 0;
        const int64_t rl = x > 0 ? regions[cur - 1] : 0;
        const int64_t rb = y > 0 ? regions[cur - W] : 0;
        const int64_t ru = z > 0 ? regions[cur - (int64_t)W * H] : 0;
        if (vc == vl && vc == vu) merge(parent, rl, ru);     /* :106-116 */
        if (vc == vl && vc == vb) merge(parent, rl, rb);
        if (vc == vu && vc == vb) merge(parent, ru, rb);
        int64_t cand = INT64_MAX;                             /* :118-133 */
        if (vc == vl && rl < cand) cand = rl;
        if (vc == vb && rb < cand) cand = rb;
        if (vc == vu && ru < cand) cand = ru;
        if (cand == INT64_MAX) { parent[nreg] = -1; cand = nreg++; }
        regions[cur] = cand;
      }
  for (int64_t i = 0; i < size; i++)
    out[i] = find_root(parent, regions[i]) == 0 ? 0.0f : 1.0f;   /* GPU semantics */
  free(regions); free(parent);
}

void fill_inside_voxels_f32(const float* grid, float* out, int N, int D, int H, int W) {
  const int64_t size = (int64_t)D * H * W;
  for (int n = 0; n < N; n++) fill_one(grid + n * size, out + n * size, D, H, W);
}
