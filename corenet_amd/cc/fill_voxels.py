"""Drop-in for `corenet.cc.fill_voxels` (cc/fill_voxels.py:61-107, module.cc:18-29).

`fill_inside_voxels_gpu(grid, inplace=False)` keeps the reference's ownership
and error behaviour (fill_voxels_gpu.cu:136-171): rank-4 CUDA tensor or
ValueError; `inplace=True` mutates and returns the caller's tensor, otherwise a
fresh contiguous tensor is returned.  The call is asynchronous on the current
stream, like the reference op.
`fill_inside_voxels_cpu(grid)` (fill_voxels_cpu.cc:158-183) is the library's own
host twin (`crn_fill_voxels_cpu`, csrc/fill_voxels_cpu.cpp): CPU tensor in, fresh
contiguous CPU tensor out, reference CPU semantics (only voxels that are not
outside are overwritten with 1).  It is a separate operator of the
boundary, not a fallback: the GPU op never routes through it.
"""
from __future__ import annotations

import torch as t

from corenet_amd.backend import default_backend, _DTYPE_CODE


def get_module(verbose=False):
  """cc/fill_voxels.py:61-99 returned the JIT-compiled torch extension; here the
  AOT-built C-ABI library plays that role."""
  from corenet_amd import _lib
  return _lib.lib()


def fill_inside_voxels_cpu(grid: t.Tensor) -> t.Tensor:
  if grid.device.type != "cpu":
    raise ValueError("Only CPU tensors are supported currently")       # fill_voxels_cpu.cc:159-161
  if grid.dim() != 4:
    raise ValueError("Expecting rank 4 tensor")                         # :163-166
  if grid.dtype not in _DTYPE_CODE:
    raise ValueError(f"unsupported dtype {grid.dtype}")                 # AT_DISPATCH_ALL_TYPES
  out = grid.clone(memory_format=t.contiguous_format)                   # :168
  n, d, h, w = out.shape
  if out.numel():
    from corenet_amd import _lib
    _lib.lib().crn_fill_voxels_cpu(out.data_ptr(), out.data_ptr(), _DTYPE_CODE[grid.dtype], n, d, h, w, 0)
  return out


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
