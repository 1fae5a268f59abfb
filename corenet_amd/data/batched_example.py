"""The voxelization half of `corenet.data.batched_example.voxelize`
(batched_example.py:121-197): per-mesh view->voxel matrices, surface
voxelization, flood fill, per-scene label merge -- all on the GPU, no host
round trips (the reference crosses the host<->device boundary three times)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch as t

from corenet_amd.backend import default_backend
from corenet_amd.cc import fill_voxels
from corenet_amd.geometry import voxelization


def view2voxel_matrices(grid_sampling_offset: t.Tensor, resolution) -> t.Tensor:
  """batched_example.py:153-163: translate(offset-0.5) @ scale(m,m,m), m=max(D,H,W)."""
  m = float(max(resolution))
  B = grid_sampling_offset.shape[0]
  mat = t.zeros(B, 4, 4, dtype=t.float32)
  mat[:, 0, 0] = mat[:, 1, 1] = mat[:, 2, 2] = m
  mat[:, 3, 3] = 1.0
  mat[:, :3, 3] = grid_sampling_offset.cpu().to(t.float32) - 0.5
  return mat


def voxelize(vertices: t.Tensor, mesh_num_tri: List[t.Tensor], mesh_labels: List[Sequence[int]],
             grid_sampling_offset: t.Tensor, resolution: Tuple[int, int, int],
             sub_grid_sampling: bool = False, conservative_rasterization: bool = False,
             image_resolution_multiplier=4, projection_depth_multiplier: int = 1,
             fill_inside: bool = True) -> t.Tensor:
  """Returns the label grid int32[batch, D, H, W] (batched_example.py:165-197)."""
  d, h, w = resolution
  B = len(mesh_num_tri)
  w2x = view2voxel_matrices(grid_sampling_offset, resolution)
  num = [len(v) for v in mesh_num_tri]
  mesh_v2x = t.cat([w2x[b:b + 1].expand(n, 4, 4) for b, n in enumerate(num)], 0)
  grids = voxelization.voxelize_mesh(
      vertices, t.cat([t.as_tensor(v, dtype=t.int32) for v in mesh_num_tri]), resolution, mesh_v2x,
      sub_grid_sampling=sub_grid_sampling, image_resolution_multiplier=image_resolution_multiplier,
      conservative_rasterization=conservative_rasterization,
      projection_depth_multiplier=projection_depth_multiplier)
  if fill_inside:
    fill_voxels.fill_inside_voxels_gpu(grids, inplace=True)
  dev = grids.device
  start = t.tensor([0] + list(t.tensor(num).cumsum(0)), dtype=t.int32, device=dev)
  labels = t.tensor([int(l) for ls in mesh_labels for l in ls], dtype=t.int32, device=dev)
  out = t.empty(B, d, h, w, dtype=t.int32, device=dev)
  default_backend().merge_labels(grids, start, labels, B, d, h, w, sub_grid_sampling, out)
  return out
