"""Batching + ground-truth voxelization (`corenet.data.batched_example`, batched_example.py:30-197)
with the geometry on the GPU end to end: `batch` uploads the object-space triangles once and applies
the per-mesh object->view matrices in a HIP kernel (the reference transforms on the host inside the
DataLoader collate); `voxelize` runs surface voxelization, flood fill and the per-scene label merge
without leaving the device (the reference crosses the host<->device boundary three times)."""
from __future__ import annotations

import dataclasses
from typing import Callable, List, Optional, Sequence, Tuple

import torch as t

from corenet_amd.backend import default_backend
from corenet_amd.cc import fill_voxels
from corenet_amd.data import dataset
from corenet_amd.data.scene import TensorContainerMixin
from corenet_amd.geometry import voxelization


@dataclasses.dataclass(frozen=True)
class BatchedExample(TensorContainerMixin):
  """A batched training/evaluation example (batched_example.py:32-65)."""
  vertices: t.Tensor                         # float32[num_total_triangles, 3, 3], view space
  view_transform: t.Tensor                   # float32[batch_size, 4, 4]
  camera_transform: t.Tensor                 # float32[batch_size, 4, 4]
  mesh_num_tri: List[t.Tensor]               # List[int32[num_meshes]]
  mesh_labels: List[t.Tensor]                # List[int32[num_meshes]]
  input_image: t.Tensor                      # uint8[batch_size, 3, height, width]
  scene_id: List[str]
  grid_sampling_offset: t.Tensor             # float32[batch_size, 3], in [0, 1]^3
  v2x_transform: Optional[t.Tensor] = None   # float32[batch_size, 4, 4]
  grid: Optional[t.Tensor] = None            # int32[batch, depth, height, width]


def batch(examples: List[dataset.DatasetElement], device=None, backend=None) -> BatchedExample:
  """Batches a list of examples (batched_example.py:68-95).  The 4x4 products view . object_to_world stay
  on the host (a few dozen matrices); the vertices go to `device` untransformed and crn_transform_meshes
  applies them there.  Everything that feeds the GPU (`vertices`, transforms, image, offsets) is returned
  on `device`; the per-mesh int lists stay on the host like in the reference."""
  with t.no_grad():
    be = backend or default_backend()
    dev = t.device(device if device is not None else "cuda")
    mats, num_tri = [], []
    for ex in examples:
      num_tri.append(ex.mesh_num_tri)
      mats.append(t.matmul(ex.view_transform[None], ex.o2w_transforms))      # o2v = w2v . o2w  (:78)
    mats = t.cat(mats, 0).to(t.float32).contiguous()
    raw = t.cat([ex.mesh_vertices for ex in examples], 0).to(dev).contiguous()
    tri_mesh = (voxelization.dynamic_tile(t.cat(num_tri, 0), dev, total=int(raw.shape[0])) if dev.type == "cuda"
                else voxelization.dynamic_tile(t.cat(num_tri, 0)))
    assert tri_mesh.shape[0] == raw.shape[0] and raw.shape[1:] == (3, 3) and raw.dtype == t.float32
    all_vertices = t.empty_like(raw)
    be.transform_meshes(raw, tri_mesh, mats.to(dev), all_vertices)
    return BatchedExample(
        vertices=all_vertices,
        view_transform=t.stack([e.view_transform for e in examples], 0).to(dev),
        camera_transform=t.stack([e.camera_transform for e in examples], 0).to(dev),
        mesh_num_tri=num_tri,
        mesh_labels=[e.mesh_labels for e in examples],
        input_image=t.stack([e.input_image for e in examples], 0).to(dev),
        scene_id=[e.scene_id for e in examples],
        grid_sampling_offset=all_vertices.new_ones([len(num_tri), 3]) * 0.5)


def voxel_content_mesh_index(batch_idx: int, mesh_idx: int) -> int:
  """Sets the voxel content to the mesh index (batched_example.py:98-101)."""
  return mesh_idx + 1


def voxel_content_1(batch_idx: int, mesh_idx: int) -> int:
  """Sets the voxel content to 1 (batched_example.py:104-108)."""
  return 1


class VoxelContentSemanticLabel:
  """Sets the voxel content to the mesh semantic class (batched_example.py:111-118)."""

  def __init__(self, semantic_labels):
    self.semantic_labels = semantic_labels

  def __call__(self, batch_idx: int, mesh_idx: int) -> int:
    return self.semantic_labels[batch_idx][mesh_idx]


def view2voxel_matrices(grid_sampling_offset: t.Tensor, resolution) -> t.Tensor:
  """batched_example.py:153-163: translate(offset-0.5) @ scale(m,m,m), m=max(D,H,W)."""
  m = float(max(resolution))
  B = grid_sampling_offset.shape[0]
  mat = t.zeros(B, 4, 4, dtype=t.float32)
  mat[:, 0, 0] = mat[:, 1, 1] = mat[:, 2, 2] = m
  mat[:, 3, 3] = 1.0
  mat[:, :3, 3] = grid_sampling_offset.cpu().to(t.float32) - 0.5
  return mat


def voxelize_labels(vertices: t.Tensor, mesh_num_tri: List[t.Tensor], mesh_labels: List[Sequence[int]],
                    grid_sampling_offset: t.Tensor, resolution: Tuple[int, int, int],
                    sub_grid_sampling: bool = False, conservative_rasterization: bool = False,
                    image_resolution_multiplier=4, projection_depth_multiplier: int = 1,
                    fill_inside: bool = True) -> t.Tensor:
  """The device half of `voxelize`: label grid int32[batch, D, H, W] (batched_example.py:165-197)."""
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


def voxelize(ex: BatchedExample, resolution: Tuple[int, int, int],
             voxel_content_fn: Callable[[int, int], int] = voxel_content_mesh_index,
             sub_grid_sampling: bool = False, conservative_rasterization: bool = False,
             image_resolution_multiplier=4, projection_depth_multiplier: int = 1,
             fill_inside: bool = True) -> BatchedExample:
  """Voxelizes the batch geometry (batched_example.py:121-197): returns the batch with `grid`
  (int32[batch, D, H, W], voxel content chosen by `voxel_content_fn(batch_idx, mesh_idx)`) and the
  unshifted world->voxel transform `v2x_transform` = scale(m, m, m)."""
  with t.no_grad():
    m = float(max(resolution))
    B = ex.grid_sampling_offset.shape[0]
    labels = [[int(voxel_content_fn(b, i)) for i in range(len(nt))] for b, nt in enumerate(ex.mesh_num_tri)]
    grid = voxelize_labels(ex.vertices, ex.mesh_num_tri, labels, ex.grid_sampling_offset, resolution,
                           sub_grid_sampling=sub_grid_sampling,
                           conservative_rasterization=conservative_rasterization,
                           image_resolution_multiplier=image_resolution_multiplier,
                           projection_depth_multiplier=projection_depth_multiplier, fill_inside=fill_inside)
    v2x = t.diag(t.tensor([m, m, m, 1.0])).expand(B, 4, 4).to(ex.grid_sampling_offset.device)
    return dataclasses.replace(ex, v2x_transform=v2x, grid=grid)
