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
  """One rendered synthetic scene (scene.py:32-76).  Field order is the reference's (positional construction).

  mesh_vertices           per mesh float32[num_triangles, 3, 3], object space
  view_transform          float32[4, 4]               world -> view
  o2w_transforms          float32[num_meshes, 4, 4]   object -> world
  camera_transform        float32[4, 4]               projection
  mesh_labels             per mesh class id (str)
  mesh_visible_fractions  float32[num_meshes]
  opengl_image            uint8[height, width, 3]     low-realism rendering
  pbrt_image              uint8[height, width, 3]     high-realism rendering
  normals, texcoords, material_ids, diffuse_colors, diffuse_texture_pngs
                          per mesh extras, filled only by load_from_npz(load_extra_fields=True)"""
  mesh_vertices: List[t.Tensor]
  view_transform: t.Tensor
  o2w_transforms: t.Tensor
  camera_transform: t.Tensor
  mesh_labels: List[Text]
  mesh_visible_fractions: t.Tensor
  opengl_image: t.Tensor
  pbrt_image: t.Tensor
  normals: List[t.Tensor] = d.field(default_factory=list)
  texcoords: List[t.Tensor] = d.field(default_factory=list)
  material_ids: List[t.Tensor] = d.field(default_factory=list)
  diffuse_colors: List[t.Tensor] = d.field(default_factory=list)
  diffuse_texture_pngs: List[List[bytes]] = d.field(default_factory=list)


def _decode_image(encoded) -> t.Tensor:
  """Image bytes (WebP / PNG / ... whatever PIL reads) -> uint8[height, width, 3]."""
  import PIL.Image
  with PIL.Image.open(io.BytesIO(encoded)) as im:
    return _to_tensor(np.array(im), t.uint8)


class NpzReader:
  """Typed access to the arrays of one NPZ file (scene.py:83-103)."""

  def __init__(self, path: str):
    with open(path, "rb") as fl:
      self.npz = np.load(io.BytesIO(fl.read()), allow_pickle=True)

  def tensor(self, item: str, dtype: Optional[t.dtype] = None) -> t.Tensor:
    """The array as a tensor; with `dtype` its type is checked, not converted."""
    arr = self.npz[item]
    return t.as_tensor(arr) if dtype is None else _to_tensor(arr, dtype)

  def list(self, item: str) -> List[Any]:
    arr = self.npz[item]
    assert arr.ndim == 1
    return [v for v in arr]

  def scalar(self, item: str) -> Any:
    arr = self.npz[item]
    assert arr.ndim == 0
    return arr


# extra per-mesh fields of a mesh NPZ: (Scene field, NPZ key, dtype or None for a 0-d object array)
_EXTRA_FIELDS = (("normals", "normals", t.float32), ("material_ids", "material_ids", t.int32),
                 ("texcoords", "texcoords", t.float32), ("diffuse_colors", "diffuse_colors", t.float32),
                 ("diffuse_texture_pngs", "diffuse_texture_pngs", None))


def load_from_npz(path: Text, meshes_dir: Text, load_extra_fields=False) -> Scene:
  """Reads a scene NPZ and the meshes it names (scene.py:106-151): object i is
  `<meshes_dir>/<mesh_labels[i]>/<mesh_filenames[i]>.npz`; its `vertices` always, the fields of _EXTRA_FIELDS
  (not needed by the training pipeline) only on request."""
  rd = NpzReader(path)
  labels = [str(v) for v in rd.list("mesh_labels")]
  names = [str(v) for v in rd.list("mesh_filenames")]
  sc = Scene(mesh_vertices=[],
             view_transform=rd.tensor("view_transform", t.float32),
             o2w_transforms=rd.tensor("mesh_object_to_world_transforms", t.float32),
             camera_transform=rd.tensor("camera_transform", t.float32),
             mesh_labels=labels,
             mesh_visible_fractions=rd.tensor("mesh_visible_fractions", t.float32),
             opengl_image=_decode_image(rd.scalar("opengl_image")),
             pbrt_image=_decode_image(rd.scalar("pbrt_image")))
  for label, name in zip(labels, names):
    mesh = NpzReader(os.path.join(meshes_dir, label, name) + ".npz")
    sc.mesh_vertices.append(mesh.tensor("vertices", t.float32))
    if load_extra_fields:
      for field, key, dtype in _EXTRA_FIELDS:
        getattr(sc, field).append(mesh.scalar(key) if dtype is None else mesh.tensor(key, dtype))
  return sc
