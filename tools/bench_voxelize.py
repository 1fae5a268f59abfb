#!/usr/bin/env python
"""Ground-truth side of one training batch: 12 UV-sphere meshes of ~20k triangles each (B=4 triplets),
128^3 grid, image_resolution_multiplier 8: surface voxelization, flood fill, label merge, and the three together
(`batched_example.voxelize_labels`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as t
from corenet_amd.backend import default_backend
from corenet_amd.data import batched_example as B
from corenet_amd.geometry import voxelization

def uv_sphere(nlat, nlon, c, r):
  th = np.linspace(0, np.pi, nlat + 1); ph = np.linspace(0, 2 * np.pi, nlon + 1)
  p = lambda i, j: c + r * np.array([np.sin(th[i]) * np.cos(ph[j]), np.sin(th[i]) * np.sin(ph[j]), np.cos(th[i])])
  tris = []
  for i in range(nlat):
    for j in range(nlon):
      a, b, cc, d = p(i, j), p(i + 1, j), p(i + 1, j + 1), p(i, j + 1)
      tris += [[a, b, cc], [a, cc, d]]
  return np.array(tris, np.float32)

rng = np.random.RandomState(0)
meshes = [uv_sphere(100, 100, 0.3 + 0.4 * rng.rand(3), 0.08 + 0.1 * rng.rand()) for _ in range(12)]
tris = t.tensor(np.concatenate(meshes)).cuda()
nt = [t.tensor([len(m) for m in meshes[3 * b:3 * b + 3]], dtype=t.int32) for b in range(4)]
labels = [[1, 2, 3]] * 4
off = t.full((4, 3), 0.5)
def timeit(fn, n=10):
  for _ in range(2): fn()
  t.cuda.synchronize(); a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); t.cuda.synchronize(); return a.elapsed_time(b) / n
v2x = B.view2voxel_matrices(off, (128,) * 3)
mv = t.cat([v2x[b:b + 1].expand(3, 4, 4) for b in range(4)])
print(f"{tris.shape[0]} triangles, 12 meshes")
ms = timeit(lambda: voxelization.voxelize_mesh(tris, t.cat(nt), (128,) * 3, mv, image_resolution_multiplier=8))
print(f"voxelize_mesh 12 x 128^3, multiplier 8: {ms * 1e3:.0f} us")
ms = timeit(lambda: B.voxelize_labels(tris, nt, labels, off, (128,) * 3, image_resolution_multiplier=8))
print(f"voxelize_labels (raster + fill + merge): {ms * 1e3:.0f} us")
g = B.voxelize_labels(tris, nt, labels, off, (128,) * 3, image_resolution_multiplier=8)
print("label histogram:", t.bincount(g.reshape(-1).long()).tolist())
from corenet_amd.cc import fill_voxels
grids = voxelization.voxelize_mesh(tris, t.cat(nt), (128,) * 3, mv, image_resolution_multiplier=8)
print(f"fill_inside_voxels_gpu (in place, 12 x 128^3 sphere surfaces): {timeit(lambda: fill_voxels.fill_inside_voxels_gpu(grids, inplace=True)) * 1e3:.0f} us")
be = default_backend()
start = t.tensor([0, 3, 6, 9, 12], dtype=t.int32).cuda(); lab = t.tensor([1, 2, 3] * 4, dtype=t.int32).cuda()
out = t.empty(4, 128, 128, 128, dtype=t.int32).cuda()
print(f"merge_labels: {timeit(lambda: be.merge_labels(grids, start, lab, 4, 128, 128, 128, False, out)) * 1e3:.0f} us")
import time
t.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): B.view2voxel_matrices(off, (128,) * 3)
print(f"view2voxel_matrices host: {(time.perf_counter() - t0) / 10 * 1e6:.0f} us")
t0 = time.perf_counter()
for _ in range(10): voxelization.dynamic_tile(t.cat(nt)).cuda()
t.cuda.synchronize(); print(f"dynamic_tile + upload: {(time.perf_counter() - t0) / 10 * 1e6:.0f} us")
