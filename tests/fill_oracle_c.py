"""ctypes loader of the C fill oracle (oracle/fill_voxels_oracle.c); test-only."""
import ctypes
import os
import subprocess

import numpy as np

_ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_SO = os.path.join(_ORACLE, "_build", "libfill_oracle.so")


def _lib():
  if not os.path.exists(_SO):
    subprocess.check_call(["make", "-C", _ORACLE, "-s"])
  lib = ctypes.CDLL(_SO)
  lib.fill_inside_voxels_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4
  return lib


def fill(grid: np.ndarray) -> np.ndarray:
  g = np.ascontiguousarray(grid, np.float32)
  out = np.empty_like(g)
  N, D, H, W = g.shape
  _lib().fill_inside_voxels_f32(g.ctypes.data, out.ctypes.data, N, D, H, W)
  return out.astype(grid.dtype)
