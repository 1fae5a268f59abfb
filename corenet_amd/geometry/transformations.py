"""Homogeneous transforms used on the hot path's host side (geometry/transformations.py:25-58, 201-262).
Only what the model / voxelizer / super-resolution callers and the benchmark's camera need; plain torch host math."""
import math

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


def look_at_rh(eye, center, up) -> t.Tensor:
  """Right-handed world->view matrix f32[4, 4] looking from `eye` at `center` (transformations.py:201-220):
  rows = (side, up', -forward) with the eye moved to the origin."""
  eye, center, up = (t.as_tensor(v, dtype=t.float32) for v in (eye, center, up))
  fwd = center - eye
  fwd = fwd / fwd.norm()
  side = t.linalg.cross(fwd, up)
  side = side / side.norm()
  rot = t.stack([side, t.linalg.cross(side, fwd), -fwd])
  m = t.eye(4, dtype=t.float32)
  m[:3, :3] = rot
  m[:3, 3] = -(rot @ eye)
  return m


def perspective_rh(fov_y: float, aspect: float, z_near: float, z_far: float) -> t.Tensor:
  """Right-handed OpenGL perspective projection f32[4, 4], depth to [-1, 1] (transformations.py:244-262)."""
  th = math.tan(fov_y / 2)
  m = t.zeros(4, 4, dtype=t.float32)
  m[0, 0] = 1.0 / (aspect * th)
  m[1, 1] = 1.0 / th
  m[2, 2] = -(z_far + z_near) / (z_far - z_near)
  m[2, 3] = -(2 * z_far * z_near) / (z_far - z_near)
  m[3, 2] = -1.0
  return m
