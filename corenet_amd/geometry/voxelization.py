"""Drop-in for `corenet.geometry.voxelization` (voxelization.py:32-182): the GL
geometry+fragment shader pipeline is replaced by the HIP software rasterizer
crn_voxelize_mesh; the result stays on the GPU (no host round trip)."""
from __future__ import annotations

from typing import Tuple

import torch as t

from corenet_amd.backend import default_backend


def dynamic_tile(partition_lengths: t.Tensor, device=None, total: int = None) -> t.Tensor:
  """misc_util.py:32-48: [n0 zeros, n1 ones, ...] int32.  With `device` the table is expanded there (only the
  few partition lengths cross the bus; `total` = sum of the lengths avoids a device->host read-back)."""
  if device is None:
    return t.repeat_interleave(t.arange(len(partition_lengths), dtype=t.int32),
                               partition_lengths.to(t.int64).cpu())
  n = partition_lengths.to(device=device, dtype=t.int64, non_blocking=True)
  return t.repeat_interleave(t.arange(len(partition_lengths), dtype=t.int32, device=device), n, output_size=total)


def voxelize_mesh(triangles, mesh_num_tri, resolution: Tuple[int, int, int], view2voxel,
                  sub_grid_sampling: bool = False, image_resolution_multiplier: float = 4,
                  conservative_rasterization: bool = False, projection_depth_multiplier: int = 1,
                  cuda_device=None) -> t.Tensor:
  """voxelization.py:32-164.  Returns float32[num_meshes, D, H, W] (or the
  (2D+1,2H+1,2W+1) sub-grid) on the GPU."""
  dev = t.device("cuda" if cuda_device is None else f"cuda:{cuda_device}")
  triangles = t.as_tensor(triangles, dtype=t.float32)
  assert triangles.shape[1:] == (3, 3)
  mesh_num_tri = t.as_tensor(mesh_num_tri, dtype=t.int32)
  assert mesh_num_tri.dim() == 1
  view2voxel = t.as_tensor(view2voxel, dtype=t.float32)
  M = len(mesh_num_tri)
  if view2voxel.dim() == 2:
    view2voxel = view2voxel[None].expand(M, 4, 4)
  assert view2voxel.shape == (M, 4, 4)
  if sub_grid_sampling and image_resolution_multiplier % 2 == 0:
    raise ValueError(
        "image_resolution_multiplier must be off if sub_grid_sampling is True")   # :107-109
  if sub_grid_sampling and projection_depth_multiplier == 0:
    raise ValueError("projection_depth_multiplier must be 1 if sub_grid_sampling is True")
  D, H, W = resolution
  tri = triangles.to(dev).contiguous()
  if not mesh_num_tri.is_cuda:
    assert int(mesh_num_tri.sum()) == triangles.shape[0], "mesh_num_tri must add up to the number of triangles"
  tri_mesh = dynamic_tile(mesh_num_tri, dev, total=int(triangles.shape[0]))
  v2v = view2voxel.to(dev).contiguous()
  shape = (M, 2 * D + 1, 2 * H + 1, 2 * W + 1) if sub_grid_sampling else (M, D, H, W)
  grid = t.empty(shape, dtype=t.float32, device=dev)
  default_backend().voxelize_mesh(tri, tri_mesh, v2v, M, D, H, W,
                                  int(image_resolution_multiplier) if sub_grid_sampling else 0,
                                  image_resolution_multiplier, conservative_rasterization,
                                  projection_depth_multiplier, grid)
  return grid


def get_sub_grid_centers(grid: t.Tensor) -> t.Tensor:
  """voxelization.py:167-182 (a strided view, no copy)."""
  return grid[:, 1::2, 1::2, 1::2]
