"""Dataset over scene NPZ files (`corenet.data.dataset`, dataset.py:39-252): same element type,
class-index mapping, slicing / shuffling / concatenation semantics.  The JSON config is read with
plain dataclasses (the reference's jsonschema machinery is out of scope)."""
from __future__ import annotations

import dataclasses
import json
import math
import os
from typing import Callable, Iterable, List, Mapping, Optional, Text, Tuple, Union

import numpy as np
import torch as t
import torch.utils.data

from corenet_amd.data import scene

VOID_LABEL_NAME = "__void__"       # dataset.py:39


@dataclasses.dataclass
class DatasetClass:
  id: Text
  human_readable: Text


@dataclasses.dataclass
class DatasetConfig:
  """dataset.py:48-51; `from_dict` takes the JSON written by the reference's tools."""
  classes: List[DatasetClass]
  files: List[Text]

  @classmethod
  def from_dict(cls, v: Mapping) -> "DatasetConfig":
    return cls(classes=[DatasetClass(id=c["id"], human_readable=c["human_readable"]) for c in v["classes"]],
               files=list(v["files"]))


@dataclasses.dataclass
class DatasetElement(scene.TensorContainerMixin):
  """A single dataset element (dataset.py:54-82)."""
  scene_id: str
  mesh_vertices: t.Tensor        # float32[num_total_tri, 3, 3], untransformed, all meshes
  mesh_num_tri: t.Tensor         # int32[num_meshes]
  view_transform: t.Tensor       # float32[4, 4]
  camera_transform: t.Tensor     # float32[4, 4]
  o2w_transforms: t.Tensor       # float32[num_meshes, 4, 4]
  mesh_labels: t.Tensor          # int32[num_meshes]
  input_image: t.Tensor          # uint8[3, height, width]


PipelineTransformation = Callable[[scene.Scene, DatasetElement], DatasetElement]
PipelineTransformations = Optional[List[PipelineTransformation]]


def to_dataset_element(ex: scene.Scene, file_name: str, class_to_int_mapping: Mapping[str, int],
                       high_realism: bool) -> DatasetElement:
  """Converts a scene to a dataset element (dataset.py:89-115)."""
  image = ex.pbrt_image if high_realism else ex.opengl_image
  input_image = scene._to_tensor(image, t.uint8).permute([2, 0, 1])
  mesh_labels = t.as_tensor([int(class_to_int_mapping[v]) for v in ex.mesh_labels], dtype=t.int32)
  mesh_num_tri = t.as_tensor([v.shape[0] for v in ex.mesh_vertices], dtype=t.int32)
  return DatasetElement(
      scene_id=os.path.splitext(file_name)[0], mesh_vertices=t.cat(ex.mesh_vertices, dim=0),
      mesh_num_tri=mesh_num_tri, view_transform=ex.view_transform, camera_transform=ex.camera_transform,
      o2w_transforms=ex.o2w_transforms, mesh_labels=mesh_labels.view(-1), input_image=input_image)


def build_class_structures(dataset_config: DatasetConfig) -> Tuple[Tuple[str, ...], Mapping[str, int]]:
  """Class names sorted by human-readable name with "__void__" first; class id -> index (dataset.py:118-143)."""
  sorted_classes = sorted(dataset_config.classes, key=lambda v: v.human_readable)
  classes = tuple([VOID_LABEL_NAME] + [v.human_readable for v in sorted_classes])
  class_to_int_mapping = {v.id: i + 1 for i, v in enumerate(sorted_classes)}     # 0 is reserved for empty/void
  if len(class_to_int_mapping) != len(set(class_to_int_mapping.values())):
    raise ValueError("Found duplicate class IDs")
  return classes, class_to_int_mapping


class CoReNetDatasetImpl(torch.utils.data.Dataset):
  """A dataset on disk: a JSON DatasetConfig next to its scene NPZ files (dataset.py:146-196)."""

  def __init__(self, dataset_path: Text, meshes_dir: Text, high_realism: bool = True,
               data_transforms: PipelineTransformations = None):
    self.high_realism = high_realism
    self.data_transforms = data_transforms or []
    self.dataset_path = dataset_path
    self.meshes_dir = meshes_dir
    with open(dataset_path, "r") as fl:
      dataset_config = DatasetConfig.from_dict(json.load(fl))
    self.root_directory = os.path.dirname(self.dataset_path)
    classes, self.class_to_int_mapping = build_class_structures(dataset_config)
    # numpy arrays instead of lists: no copy-on-access growth in DataLoader workers (dataset.py:174-180)
    self.files = np.array(dataset_config.files)
    self.classes = np.array(classes)

  def __getitem__(self, index: int) -> DatasetElement:
    file_name = str(self.files[index])
    inex = scene.load_from_npz(os.path.join(self.root_directory, file_name), self.meshes_dir,
                               load_extra_fields=False)
    dex = to_dataset_element(inex, file_name, self.class_to_int_mapping, self.high_realism)
    for transf in self.data_transforms:
      dex = transf(inex, dex)
    return dex

  def __len__(self) -> int:
    return self.files.shape[0]


class CoReNetDataset(torch.utils.data.Dataset):
  """Virtual dataset: slicing, fractions, shuffling, concatenation with the class list kept
  (dataset.py:199-241)."""

  def __init__(self, d: torch.utils.data.Dataset, classes: Union[np.ndarray, Tuple[str, ...]],
               indices: Optional[t.Tensor] = None):
    self._dataset = d
    self.classes = np.array(classes)
    if indices is None:
      indices = t.arange(len(d), device="cpu")
    self.indices = indices

  def __add__(self, other: "CoReNetDataset") -> "CoReNetDataset":
    if not np.array_equal(other.classes, self.classes):
      raise ValueError("The classes of both datasets must match.")
    return concatenate([self, other])

  def __len__(self):
    return self.indices.shape[0]

  def __getitem__(self, index: Union[int, slice]) -> Union[DatasetElement, "CoReNetDataset"]:
    if isinstance(index, slice):
      return CoReNetDataset(self._dataset, self.classes, self.indices[index])
    return self._dataset[int(self.indices[index])]

  def take_fraction(self, start: float, end: float) -> "CoReNetDataset":
    assert 0 <= start <= end <= 1
    return self[int(math.floor(start * len(self))): int(math.ceil(end * len(self)))]

  def shuffle(self, seed: int) -> "CoReNetDataset":
    g = t.Generator()
    g.manual_seed(seed)
    indices = torch.randperm(self.indices.shape[0], generator=g, device="cpu")
    return CoReNetDataset(self._dataset, self.classes, self.indices[indices])


def concatenate(datasets: Iterable[CoReNetDataset]) -> CoReNetDataset:
  """dataset.py:244-252."""
  datasets = list(datasets)
  if len(datasets) == 1:
    return datasets[0]
  all_classes = np.array([v.classes for v in datasets])
  assert (all_classes[0:1] == all_classes).all()
  return CoReNetDataset(torch.utils.data.ConcatDataset(datasets), all_classes[0])
