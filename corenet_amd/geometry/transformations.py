"""Homogeneous transforms used on the hot path's host side (geometry/transformations.py:25-58).
Only what the model / voxelizer / super-resolution callers need; plain torch host math."""
import torch as t


def scale(v) -> t.Tensor:
  """Homogeneous scale matrix f32[N+1, N+1] from the scale vector v[N] (transformations.py:25-37)."""
  v = t.as_tensor(v, dtype=t.float32)
  return t.diag(t.cat([v, v.new_ones([1])], dim=0))


def translate(v) -> t.Tensor:
  """Homogeneous translation matrix f32[..., N+1, N+1] from v[..., N] (transformations.py:40-58)."""
  v = t.as_tensor(v, dtype=t.float32)
  n = v.shape[-1]
  result = t.eye(n + 1, dtype=t.float32, device=v.device).expand(v.shape[:-1] + (n + 1, n + 1)).clone()
  result[..., :n, n] = v
  return result
