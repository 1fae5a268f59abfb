"""Drop-in for `corenet.cc.fill_voxels` (cc/fill_voxels.py:61-107, module.cc:18-29).

`fill_inside_voxels_gpu(grid, inplace=False)` keeps the reference's ownership
and error behaviour (fill_voxels_gpu.cu:136-171): rank-4 CUDA tensor or
ValueError; `inplace=True` mutates and returns the caller's tensor, otherwise a
fresh contiguous tensor is returned.  There is deliberately no
`fill_inside_voxels_cpu` product path: the CPU twin lives in oracle/ as test
infrastructure.
"""
from __future__ import annotations

import torch as t

from corenet_amd.backend import default_backend, _DTYPE_CODE


def get_module(verbose=False):
  """cc/fill_voxels.py:61-99 returned the JIT-compiled torch extension; here the
  AOT-built C-ABI library plays that role."""
  from corenet_amd import _lib
  return _lib.lib()


def fill_inside_voxels_gpu(grid: t.Tensor, inplace: bool = False) -> t.Tensor:
  if not grid.is_cuda:
    raise ValueError("Only CUDA tensors are supported by this OP")     # fill_voxels_gpu.cu:137-139
  if grid.dim() != 4:
    raise ValueError("Expecting rank 4 tensor")                         # :141-144
  if grid.dtype not in _DTYPE_CODE:
    raise ValueError(f"unsupported dtype {grid.dtype}")                 # AT_DISPATCH_ALL_TYPES
  src = grid if grid.is_contiguous() else grid.contiguous()
  if inplace and src is grid:
    out = grid
  else:
    out = t.empty_like(src)
  default_backend().fill_voxels(src, out)
  if inplace and out is not grid:
    grid.copy_(out)
    return grid
  return out
