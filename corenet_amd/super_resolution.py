"""Drop-in for `corenet.super_resolution` (super_resolution.py:28-129): inference at a multiple of the
native grid resolution by evaluating the network at m^3 sub-voxel sampling offsets and interleaving
the results.

Same classes, argument meaning and errors as the reference.  The MI355X-specific part is
`super_resolution_from_state`: its inference function runs the offset-independent ResNet-50 encoder
once per image batch and only the decoder per offset (`CoreNet.multi_offset_pmf`), and the softmax +
stack + reshape + permute + reshape of super_resolution.py:105-112,124 is one HIP kernel with coalesced
writes (`crn_softmax_superres`).
"""
from typing import Tuple

import torch as t

from corenet_amd.geometry import transformations


class MultiOffsetInferenceFn:
  def __call__(self, input_image: t.Tensor, camera_transform: t.Tensor, view_to_voxel_transform: t.Tensor,
               grid_offsets: t.Tensor) -> t.Tensor:
    """input_image uint8/float[B,3,h,w], camera_transform f32[B,4,4], view_to_voxel_transform f32[B,4,4],
    grid_offsets f32[num_offsets,B,3] -> pmf f32[num_offsets,B,C,D,H,W] (super_resolution.py:29-44)."""
    raise NotImplementedError()


class SuperResolutionInference:
  """super_resolution.py:46-112."""

  def __init__(self, inference_fn: MultiOffsetInferenceFn, resolution: Tuple[int, int, int]):
    self.resolution = resolution
    self.inference_fn = inference_fn
    self.offset_cache = {}

  def get_resolution_multiplier(self, output_resolution: Tuple[int, int, int]) -> int:
    """Multiplier between the native and the output resolution (super_resolution.py:53-64)."""
    rm = (t.as_tensor(output_resolution, dtype=t.float32) / t.as_tensor(self.resolution, dtype=t.float32))
    if (rm.floor() != rm.ceil()).any() or (rm < 1).any() or rm.min() != rm.max():
      raise ValueError("The output resolution should be divisible by the native resolution")
    return int(rm[0])

  def get_native_offsets(self, output_resolution: Tuple[int, int, int], grid_offsets: t.Tensor) -> t.Tensor:
    """Sampling offsets in the native grid, f32[m^3, B, 3] (super_resolution.py:66-90): offset index
    n = (iz*m + iy)*m + ix  <->  ((ix, iy, iz) + grid_offset) / m."""
    output_resolution = tuple(output_resolution)
    assert len(output_resolution) == 3
    m = self.get_resolution_multiplier(output_resolution)
    if output_resolution not in self.offset_cache:
      zz, yy, xx = t.meshgrid([t.arange(m, device="cpu")] * 3, indexing="ij")
      offsets = t.stack([xx, yy, zz], -1) / m
      self.offset_cache[output_resolution] = offsets.reshape([-1, 3])
    offsets = self.offset_cache[output_resolution].to(grid_offsets.device)
    return offsets[:, None] + grid_offsets[None, :] / m

  def __call__(self, input_image: t.Tensor, camera_transform: t.Tensor, view_to_voxel_transform: t.Tensor,
               grid_offsets: t.Tensor, output_resolution: Tuple[int, int, int]) -> t.Tensor:
    native_offsets = self.get_native_offsets(output_resolution, grid_offsets)
    m = self.get_resolution_multiplier(output_resolution)
    batch_size = input_image.shape[0]
    scale = transformations.scale([1 / m, 1 / m, 1 / m])
    view_to_voxel_transform = view_to_voxel_transform @ scale.to(view_to_voxel_transform.device)
    fused = getattr(self.inference_fn, "interleaved", None)
    if fused is not None:       # HIP path: returns [B, C, mD, mH, mW] directly
      return fused(input_image, camera_transform, view_to_voxel_transform, native_offsets, m)
    pmfs = self.inference_fn(input_image, camera_transform, view_to_voxel_transform, native_offsets)
    _, _, num_channels, d, h, w = pmfs.shape
    pmfs = pmfs.reshape([m, m, m, batch_size, num_channels, d, h, w])
    pmfs = pmfs.permute([3, 4, 5, 0, 6, 1, 7, 2])
    return pmfs.reshape([batch_size, num_channels, m * d, m * h, m * w])


class CoreNetMultiOffset(MultiOffsetInferenceFn):
  """The inference_fn of super_resolution_from_state (super_resolution.py:115-126) with encoder reuse."""

  def __init__(self, model):
    self.model = model

  def _v2s(self, camera_transform, view_to_voxel_transform):
    return camera_transform @ view_to_voxel_transform.inverse()     # super_resolution.py:121

  def __call__(self, input_image, camera_transform, view_to_voxel_transform, grid_offsets):
    return self.model.multi_offset_pmf(input_image, self._v2s(camera_transform, view_to_voxel_transform),
                                       grid_offsets)

  def interleaved(self, input_image, camera_transform, view_to_voxel_transform, grid_offsets, m):
    return self.model.multi_offset_pmf(input_image, self._v2s(camera_transform, view_to_voxel_transform),
                                       grid_offsets, resolution_multiplier=m)


def super_resolution_from_state(state) -> SuperResolutionInference:
  """state: anything with `.model` (a corenet_amd CoreNet in eval mode), like state.State."""
  model = state.model
  return SuperResolutionInference(CoreNetMultiOffset(model), tuple(model.config.decoder.resolution))
