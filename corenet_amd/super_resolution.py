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
  """Inference at m times the native grid resolution from m^3 native passes at shifted sampling offsets
  (super_resolution.py:46-112)."""

  def __init__(self, inference_fn: MultiOffsetInferenceFn, resolution: Tuple[int, int, int]):
    self.inference_fn = inference_fn
    self.resolution = resolution
    self.offset_cache = {}          # output resolution -> unit-cube offsets f32[m^3, 3]

  def get_resolution_multiplier(self, output_resolution: Tuple[int, int, int]) -> int:
    """m = output / native resolution: one integer >= 1 for all three axes, else ValueError
    (super_resolution.py:53-64)."""
    ratio = t.tensor(output_resolution, dtype=t.float32) / t.tensor(self.resolution, dtype=t.float32)
    whole = bool((ratio == ratio.round()).all())
    if not whole or float(ratio.min()) < 1 or float(ratio.min()) != float(ratio.max()):
      raise ValueError("The output resolution should be divisible by the native resolution")
    return int(ratio[0])

  def _unit_offsets(self, output_resolution: Tuple[int, int, int], m: int) -> t.Tensor:
    """(ix, iy, iz) / m for offset index n = (iz * m + iy) * m + ix, cached per output resolution."""
    hit = self.offset_cache.get(output_resolution)
    if hit is None:
      n = t.arange(m ** 3)
      hit = t.stack([n % m, (n // m) % m, n // (m * m)], 1).to(t.float32) / m
      self.offset_cache[output_resolution] = hit
    return hit

  def get_native_offsets(self, output_resolution: Tuple[int, int, int], grid_offsets: t.Tensor) -> t.Tensor:
    """Sampling offsets of the m^3 native passes, f32[m^3, B, 3]: ((ix, iy, iz) + grid_offset) / m
    (super_resolution.py:66-90)."""
    output_resolution = tuple(output_resolution)
    assert len(output_resolution) == 3
    m = self.get_resolution_multiplier(output_resolution)
    unit = self._unit_offsets(output_resolution, m).to(grid_offsets.device)
    return unit.unsqueeze(1) + grid_offsets.unsqueeze(0) / m

  def __call__(self, input_image: t.Tensor, camera_transform: t.Tensor, view_to_voxel_transform: t.Tensor,
               grid_offsets: t.Tensor, output_resolution: Tuple[int, int, int]) -> t.Tensor:
    """pmf f32[B, C, m*D, m*H, m*W] (super_resolution.py:92-112)."""
    m = self.get_resolution_multiplier(output_resolution)
    offsets = self.get_native_offsets(output_resolution, grid_offsets)
    # the native grid spans the same volume with m times fewer voxels per axis
    v2x_native = view_to_voxel_transform @ transformations.scale([1.0 / m] * 3).to(view_to_voxel_transform.device)
    fused = getattr(self.inference_fn, "interleaved", None)
    if fused is not None:       # HIP path: softmax + interleave in one kernel, [B, C, mD, mH, mW] directly
      return fused(input_image, camera_transform, v2x_native, offsets, m)
    pmfs = self.inference_fn(input_image, camera_transform, v2x_native, offsets)      # [m^3, B, C, D, H, W]
    B, C, D, H, W = pmfs.shape[1:]
    # [iz, iy, ix, B, C, D, H, W] -> [B, C, D, iz, H, iy, W, ix] -> fine voxel = coarse voxel * m + sub-position
    fine = pmfs.reshape(m, m, m, B, C, D, H, W).permute(3, 4, 5, 0, 6, 1, 7, 2)
    return fine.reshape(B, C, m * D, m * H, m * W)


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
