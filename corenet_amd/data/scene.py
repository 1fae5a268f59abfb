"""Scene NPZ reader: the host half of the data path (`corenet.data.scene`, scene.py:32-151;
on-disk format doc/data_format_and_coordinate_systems.md:9-31).  Decoding NPZ / image bytes is
host work by nature; everything after it (mesh transforms, voxelization, flood fill, label merge)
runs on the GPU (`corenet_amd.data.batched_example`).  Local file system only: the reference's
`gs://` transport (file_system.py) is out of scope."""
from __future__ import annotations

import dataclasses as d
import io
import os
from typing import Any, List, Optional, Text

import numpy as np
import torch as t


def _to_tensor(v, dtype: t.dtype) -> t.Tensor:
  """misc_util.to_tensor (misc_util.py:51-79): arrays keep their type and must already match."""
  if not t.is_tensor(v):
    v = t.as_tensor(v) if hasattr(v, "__array_interface__") else t.as_tensor(v, dtype=dtype)
  if v.dtype != dtype:
    raise ValueError(f"Expecting type '{dtype}', found '{v.dtype}'")
  return v


class TensorContainerMixin:
  """misc_util.TensorContainerMixin (misc_util.py:92-115): map over the tensors of a dataclass."""

  def _apply(self, fn):
    result = []
    for field in d.astuple(self):
      if t.is_tensor(field):
        field = fn(field)
      elif isinstance(field, (list, tuple)):
        field = [fn(e) if t.is_tensor(e) else e for e in field]
      result.append(field)
    return type(self)(*result)

  def cuda(self):
    return self._apply(lambda v: v.cuda())

  def cpu(self):
    return self._apply(lambda v: v.cpu())

  def numpy(self):
    return self._apply(lambda v: v.numpy())

  def to(self, device):
    return self._apply(lambda v: v.to(device))


@d.dataclass(frozen=True)
class Scene(TensorContainerMixin):
  """A rendered synthetic scene (scene.py:32-76)."""
  mesh_vertices: List[t.Tensor]          # List[float32[num_triangles, 3, 3]], object space
  view_transform: t.Tensor               # float32[4, 4] world -> view
  o2w_transforms: t.Tensor               # float32[num_meshes, 4, 4]
  camera_transform: t.Tensor             # float32[4, 4]
  mesh_labels: List[Text]
  mesh_visible_fractions: t.Tensor       # float32[num_meshes]
  opengl_image: t.Tensor                 # uint8[height, width, 3]
  pbrt_image: t.Tensor                   # uint8[height, width, 3]
  normals: List[t.Tensor] = d.field(default_factory=list)
  texcoords: List[t.Tensor] = d.field(default_factory=list)
  material_ids: List[t.Tensor] = d.field(default_factory=list)
  diffuse_colors: List[t.Tensor] = d.field(default_factory=list)
  diffuse_texture_pngs: List[List[bytes]] = d.field(default_factory=list)


def _load_image(i) -> t.Tensor:
  import PIL.Image
  return _to_tensor(np.array(PIL.Image.open(io.BytesIO(i))), t.uint8)


class NpzReader:
  """scene.py:83-103."""

  def __init__(self, path: str):
    with open(path, "rb") as fl:
      self.npz = np.load(io.BytesIO(fl.read()), allow_pickle=True)

  def tensor(self, item: str, dtype: Optional[t.dtype] = None) -> t.Tensor:
    result = self.npz[item]
    return _to_tensor(result, dtype) if dtype else t.as_tensor(result)

  def list(self, item: str) -> List[Any]:
    result = self.npz[item]
    assert len(result.shape) == 1
    return list(result)

  def scalar(self, item: str) -> Any:
    result = self.npz[item]
    assert len(result.shape) == 0
    return result


def load_from_npz(path: Text, meshes_dir: Text, load_extra_fields=False) -> Scene:
  """Loads one scene and the ShapeNet meshes it points to (scene.py:106-151): the mesh of object i is
  `<meshes_dir>/<mesh_labels[i]>/<mesh_filenames[i]>.npz`."""
  scene_npz = NpzReader(path)
  mesh_paths = [os.path.join(meshes_dir, *v) + ".npz"
                for v in zip(scene_npz.list("mesh_labels"), scene_npz.list("mesh_filenames"))]
  result = Scene(
      mesh_vertices=[],
      view_transform=scene_npz.tensor("view_transform", t.float32),
      o2w_transforms=scene_npz.tensor("mesh_object_to_world_transforms", t.float32),
      camera_transform=scene_npz.tensor("camera_transform", t.float32),
      mesh_labels=[v for v in scene_npz.list("mesh_labels")],
      opengl_image=_load_image(scene_npz.scalar("opengl_image")),
      pbrt_image=_load_image(scene_npz.scalar("pbrt_image")),
      mesh_visible_fractions=scene_npz.tensor("mesh_visible_fractions", t.float32))
  for mesh_path in mesh_paths:
    mesh_npz = NpzReader(mesh_path)
    result.mesh_vertices.append(mesh_npz.tensor("vertices", t.float32))
    if load_extra_fields:
      result.normals.append(mesh_npz.tensor("normals", t.float32))
      result.material_ids.append(mesh_npz.tensor("material_ids", t.int32))
      result.texcoords.append(mesh_npz.tensor("texcoords", t.float32))
      result.diffuse_colors.append(mesh_npz.tensor("diffuse_colors", t.float32))
      result.diffuse_texture_pngs.append(mesh_npz.scalar("diffuse_texture_pngs"))
  return result
