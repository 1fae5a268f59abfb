"""Datasets of scene NPZ files: the element type, the class-index mapping and the slicing / shuffling /
concatenation behaviour of `corenet.data.dataset` (dataset.py:39-252), re-implemented for this package.
The JSON description is read with plain dataclasses (the reference's jsonschema machinery is out of scope,
DESIGN section 7) and only local paths are supported."""
from __future__ import annotations

import dataclasses
import json
import math
import os
from typing import Callable, Iterable, List, Mapping, Optional, Sequence, Text, Tuple, Union

import numpy as np
import torch as t
import torch.utils.data

from corenet_amd.data import scene

VOID_LABEL_NAME = "__void__"       # class 0 of every dataset (dataset.py:39)


@dataclasses.dataclass
class DatasetClass:
  id: Text
  human_readable: Text


@dataclasses.dataclass
class DatasetConfig:
  """What the dataset JSON holds (dataset.py:48-51)."""
  classes: List[DatasetClass]
  files: List[Text]

  @classmethod
  def from_dict(cls, raw: Mapping) -> "DatasetConfig":
    return cls([DatasetClass(str(c["id"]), str(c["human_readable"])) for c in raw["classes"]],
               [str(f) for f in raw["files"]])


@dataclasses.dataclass
class DatasetElement(scene.TensorContainerMixin):
  """One scene, ready for batching (dataset.py:54-82).

  scene_id          file name of the scene without extension
  mesh_vertices     float32[total_triangles, 3, 3]  object-space triangles of all meshes, concatenated
  mesh_num_tri      int32[num_meshes]               triangles per mesh
  view_transform    float32[4, 4]                   world -> view
  camera_transform  float32[4, 4]                   view -> image (projection)
  o2w_transforms    float32[num_meshes, 4, 4]       object -> world, per mesh
  mesh_labels       int32[num_meshes]               class index per mesh (0 = void)
  input_image       uint8[3, height, width]"""
  scene_id: str
  mesh_vertices: t.Tensor
  mesh_num_tri: t.Tensor
  view_transform: t.Tensor
  camera_transform: t.Tensor
  o2w_transforms: t.Tensor
  mesh_labels: t.Tensor
  input_image: t.Tensor


PipelineTransformation = Callable[[scene.Scene, DatasetElement], DatasetElement]
PipelineTransformations = Optional[List[PipelineTransformation]]


def build_class_structures(dataset_config: DatasetConfig) -> Tuple[Tuple[str, ...], Mapping[str, int]]:
  """(class names, class id -> index): names ordered alphabetically by their human-readable form behind
  "__void__", indices counted from 1 in that order (dataset.py:118-143)."""
  ordered = sorted(dataset_config.classes, key=lambda c: c.human_readable)
  index_of = {c.id: pos for pos, c in enumerate(ordered, start=1)}
  if len(set(index_of.values())) != len(index_of):
    raise ValueError("Found duplicate class IDs")
  return (VOID_LABEL_NAME,) + tuple(c.human_readable for c in ordered), index_of


def to_dataset_element(ex: scene.Scene, file_name: str, class_to_int_mapping: Mapping[str, int],
                       high_realism: bool) -> DatasetElement:
  """Scene -> element (dataset.py:89-115): picks the high- or low-realism rendering (channels first), maps the
  mesh labels to class indices and concatenates the meshes."""
  rendering = ex.pbrt_image if high_realism else ex.opengl_image
  counts = [int(m.shape[0]) for m in ex.mesh_vertices]
  labels = [int(class_to_int_mapping[name]) for name in ex.mesh_labels]
  return DatasetElement(
      scene_id=os.path.splitext(file_name)[0],
      mesh_vertices=t.cat(list(ex.mesh_vertices), 0),
      mesh_num_tri=t.tensor(counts, dtype=t.int32),
      view_transform=ex.view_transform,
      camera_transform=ex.camera_transform,
      o2w_transforms=ex.o2w_transforms,
      mesh_labels=t.tensor(labels, dtype=t.int32).reshape(-1),
      input_image=scene._to_tensor(rendering, t.uint8).permute(2, 0, 1))


class CoReNetDatasetImpl(torch.utils.data.Dataset):
  """The scenes listed by a dataset JSON that sits next to them (dataset.py:146-196).  `meshes_dir` holds the
  ShapeNet meshes the scenes refer to; `data_transforms` are applied, in order, to every loaded element."""

  def __init__(self, dataset_path: Text, meshes_dir: Text, high_realism: bool = True,
               data_transforms: PipelineTransformations = None):
    with open(dataset_path, "r") as fl:
      config = DatasetConfig.from_dict(json.load(fl))
    names, self.class_to_int_mapping = build_class_structures(config)
    self.dataset_path, self.meshes_dir = dataset_path, meshes_dir
    self.root_directory = os.path.dirname(dataset_path)
    self.high_realism = high_realism
    self.data_transforms = list(data_transforms or [])
    # numpy arrays, not Python lists: DataLoader workers then share them without copy-on-access growth
    self.files = np.array(config.files)
    self.classes = np.array(names)

  def __len__(self) -> int:
    return int(self.files.shape[0])

  def __getitem__(self, index: int) -> DatasetElement:
    name = str(self.files[index])
    loaded = scene.load_from_npz(os.path.join(self.root_directory, name), self.meshes_dir, load_extra_fields=False)
    element = to_dataset_element(loaded, name, self.class_to_int_mapping, self.high_realism)
    for transform in self.data_transforms:
      element = transform(loaded, element)
    return element


class CoReNetDataset(torch.utils.data.Dataset):
  """A view (index list) on a dataset that keeps the class names through slicing, fractions, shuffling and
  concatenation (dataset.py:199-241)."""

  def __init__(self, d: torch.utils.data.Dataset, classes: Union[np.ndarray, Sequence[str]],
               indices: Optional[t.Tensor] = None):
    self._dataset = d
    self.classes = np.array(classes)
    self.indices = t.arange(len(d)) if indices is None else indices

  def _view(self, indices: t.Tensor) -> "CoReNetDataset":
    return CoReNetDataset(self._dataset, self.classes, indices)

  def __len__(self):
    return int(self.indices.shape[0])

  def __getitem__(self, index: Union[int, slice]) -> Union[DatasetElement, "CoReNetDataset"]:
    if isinstance(index, slice):
      return self._view(self.indices[index])
    return self._dataset[int(self.indices[index])]

  def __add__(self, other: "CoReNetDataset") -> "CoReNetDataset":
    if not np.array_equal(self.classes, other.classes):
      raise ValueError("The classes of both datasets must match.")
    return concatenate([self, other])

  def take_fraction(self, start: float, end: float) -> "CoReNetDataset":
    """Elements [floor(start * n), ceil(end * n))."""
    assert 0 <= start <= end <= 1
    n = len(self)
    return self[int(math.floor(start * n)):int(math.ceil(end * n))]

  def shuffle(self, seed: int) -> "CoReNetDataset":
    """A permutation drawn from a CPU generator seeded with `seed`."""
    gen = t.Generator()
    gen.manual_seed(seed)
    return self._view(self.indices[t.randperm(len(self), generator=gen, device="cpu")])


def concatenate(datasets: Iterable[CoReNetDataset]) -> CoReNetDataset:
  """One dataset after the other; all of them must have the same classes (dataset.py:244-252)."""
  parts = list(datasets)
  if len(parts) == 1:
    return parts[0]
  for p in parts[1:]:
    assert np.array_equal(p.classes, parts[0].classes)
  return CoReNetDataset(torch.utils.data.ConcatDataset(parts), parts[0].classes)
