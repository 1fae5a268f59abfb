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


class _CorenetCpp:
  """What `corenet_cpp` (cc/module.cc:18-29) looks like to its callers: an object with the two operators as attributes,
  `fill_inside_voxels_gpu(grid, inplace=False)` and `fill_inside_voxels_cpu(grid)`, so that
  `get_module().fill_inside_voxels_gpu(grid, inplace)` (cc/fill_voxels.py:102-107) works verbatim.  `.lib` is the C-ABI
  library behind them (corenet_amd._lib)."""
  __name__ = "corenet_cpp"

  @property
  def lib(self):
    from corenet_amd import _lib
    return _lib.lib()

  @staticmethod
  def fill_inside_voxels_gpu(grid: t.Tensor, inplace: bool = False) -> t.Tensor:
    return _fill_gpu(grid, inplace)

  @staticmethod
  def fill_inside_voxels_cpu(grid: t.Tensor) -> t.Tensor:
    return _fill_cpu(grid)


_corenet_cpp = _CorenetCpp()


def get_module(verbose=False):
  """cc/fill_voxels.py:61-99 returns the JIT-compiled torch extension `corenet_cpp`; here the AOT-built C-ABI library
  plays that role behind an object with the extension's two functions (the library is loaded on first use and a
  missing `.so` raises there: there is no other implementation to fall back to)."""
  return _corenet_cpp


def _fill_cpu(grid: t.Tensor) -> t.Tensor:
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


def _fill_gpu(grid: t.Tensor, inplace: bool = False) -> t.Tensor:
  if not grid.is_cuda:
    raise ValueError("Only CUDA tensors are supported by this OP")     # fill_voxels_gpu.cu:137-139
  if grid.dim() != 4:
    raise ValueError("Expecting rank 4 tensor")                         # :141-144
  if grid.dtype not in _DTYPE_CODE:
    raise ValueError(f"unsupported dtype {grid.dtype}")                 # AT_DISPATCH_ALL_TYPES
  # :162-163: in place the caller's tensor (whatever its strides), otherwise a fresh contiguous one; a non-contiguous
  # side goes to the kernel as a strided view (crn_fill_voxels_strided), like the reference's packed accessors
  out = grid if inplace else t.empty(grid.shape, dtype=grid.dtype, device=grid.device)
  if grid.numel():
    if inplace and _overlapping(grid):
      raise ValueError("fill_inside_voxels_gpu: in-place on a view with overlapping elements")
    default_backend().fill_voxels(grid, out)
  return out


def _overlapping(x: t.Tensor) -> bool:
  """An expanded (stride-0) dimension of extent > 1: in place every slab would write the same voxels."""
  return any(st == 0 and n > 1 for st, n in zip(x.stride(), x.shape))


def fill_inside_voxels_cpu(grid: t.Tensor) -> t.Tensor:
  return get_module().fill_inside_voxels_cpu(grid)                      # cc/fill_voxels.py:98-99


def fill_inside_voxels_gpu(grid: t.Tensor, inplace: bool = False) -> t.Tensor:
  return get_module().fill_inside_voxels_gpu(grid, inplace)             # cc/fill_voxels.py:102-107
