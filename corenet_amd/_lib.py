"""ctypes binding of libcorenet_hip.so (the C ABI declared in include/corenet_hip.h).

There is NO fallback: if the shared library is missing, or a call fails, this
raises.  PyTorch is used only for device memory (tensor.data_ptr()) and for the
current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch as t

_HERE = os.path.dirname(os.path.abspath(__file__))
TOOLS_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_build", "libcorenet_hip_tools.so")
PROBE_LIB_PATH = os.path.join(os.path.dirname(_HERE), "tools", "_build", "libcrn_probe.so")
LIB_PATH = os.environ.get("CORENET_HIP_LIB", TOOLS_LIB_PATH if os.environ.get("CRN_TOOLS_LIB") == "1" else
                          os.path.join(_HERE, "lib", "libcorenet_hip.so"))

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.c_void_p     # raw device pointers are passed as integers
vp = C.c_void_p
i64 = C.c_int64
i32 = C.c_int
f32 = C.c_float
sz = C.c_size_t


class CrnTapBoxes(C.Structure):
  """Mirror of crnTapBoxes (include/corenet_hip.h)."""
  _fields_ = [("n_groups", C.c_int32), ("c_groups", C.c_int32), ("n_box", (C.c_int8 * 6) * 8),
              ("c_box", (C.c_int8 * 6) * 8)]


class CrnView(C.Structure):
  """Mirror of crnView (include/corenet_hip.h)."""
  _fields_ = [("base", vp), ("B", C.c_int32), ("C", C.c_int32), ("D", C.c_int32),
              ("H", C.c_int32), ("W", C.c_int32), ("sB", i64), ("sC", i64),
              ("chan_off", vp), ("sD", C.c_int32), ("sH", C.c_int32), ("sW", C.c_int32)]


class CrnInTransform(C.Structure):
  _fields_ = [("scale", vp), ("shift", vp), ("pre_relu", C.c_int32), ("post_relu", C.c_int32)]


class CrnBnBwdFuse(C.Structure):
  """Mirror of crnBnBwdFuse (include/corenet_hip.h)."""
  _fields_ = [("x", vp), ("sB_x", i64), ("saved", vp), ("pre_relu", C.c_int32), ("ws", vp), ("ws_bytes", sz),
              ("dsum", vp), ("ndsum", C.c_int32), ("nparts", C.c_int32)]


class HipError(RuntimeError):
  pass


_SIGS = {
    "crn_conv_fwd": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32,
                     C.POINTER(CrnView), i32, i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_conv_fwd_bf3": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32,
                         C.POINTER(CrnView), i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_conv_fwd_bf3_slabs": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32,
                               C.POINTER(CrnView), i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_conv_fwd_bf3_slabs_bnbwd": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32,
                                     C.POINTER(CrnView), i32, i32, i32, i32, i32, i32, i32, vp, C.POINTER(CrnBnBwdFuse), vp],
    "crn_conv_fwd_bf3_slabs_stats": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32,
                                     C.POINTER(CrnView), i32, i32, i32, i32, i32, i32, i32, vp, i32, vp, sz, C.POINTER(C.c_int), vp],
    "crn_batch_renorm_bwd_apply": [vp, i64, vp, i64, i32, i32, i64, i32, vp, vp, vp, vp, vp, i64, vp, vp, i32, vp, i32,
                                   vp, sz, i32, vp],
    "crn_conv_wgrad_bf3": [C.POINTER(CrnView), C.POINTER(CrnInTransform), C.POINTER(CrnView), vp, i32,
                           i32, i32, i32, i32, i32, i32, i32, vp],
    "crn_conv_wgrad_bf3_boxes": [C.POINTER(CrnView), C.POINTER(CrnInTransform), C.POINTER(CrnView), vp, i32,
                                 i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_conv_wgrad": [C.POINTER(CrnView), C.POINTER(CrnInTransform), C.POINTER(CrnView), vp, i32,
                       i32, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_bf3_operands": [vp, vp, i32, i64, vp, vp],
    "crn_bf3_gather_image": [vp, vp, i32, vp, vp],
    "crn_convt_s2k7_fwd_bf3": [vp, i64, i32, i32, i32, i32, C.POINTER(CrnInTransform), vp, vp, vp, i64, i64, i32, vp],
    "crn_convt_s2k7_dgrad_bf3": [vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp],
    "crn_convt_s2k7_c2_fwd_bf3": [vp, i64, i32, i32, i32, i32, C.POINTER(CrnInTransform), vp, vp, vp, i64, i64, i32, vp],
    "crn_convt_s2k7_c2_dgrad_bf3": [vp, i64, i64, i32, i32, i32, i32, i32, vp, vp, i64, i32, vp],
    "crn_convt_s2k7_wgrad_bf3": [vp, i64, i32, i32, i32, i32, C.POINTER(CrnInTransform), vp, i64, i64, i32, vp, i32, i32, vp, vp],
    "crn_convt_s2k7_ximage": [vp, i64, i32, i32, i32, i32, C.POINTER(CrnInTransform), vp, sz, vp],
    "crn_splitk_defer": [i32],
    "crn_splitk_reserve": [i64, vp],
    "crn_splitk_release": [vp],
    "crn_set_deterministic": [i32],
    "crn_roctx_push": [C.c_char_p],
    "crn_roctx_pop": [],
    "crn_comm_unique_id": [vp],
    "crn_comm_init": [vp, i32, i32, C.POINTER(vp)],
    "crn_comm_destroy": [vp],
    "crn_comm_info": [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
    "crn_allreduce_f32": [vp, vp, i64, vp],
    "crn_broadcast_f32": [vp, vp, i64, i32, vp],
    "crn_conv_wgrad_1x1_bf3": [C.POINTER(CrnView), C.POINTER(CrnInTransform), C.POINTER(CrnView), vp, i32, i32, vp],
    "crn_conv_wgrad_2d_bf3": [C.POINTER(CrnView), C.POINTER(CrnInTransform), C.POINTER(CrnView), vp, i32, i32, i32, i32,
                              i32, i32, vp],
    "crn_conv2d_bf3": [C.POINTER(CrnView), C.POINTER(CrnInTransform), vp, i32, vp, i32, C.POINTER(CrnView),
                       i32, i32, i32, i32, i32, vp],
    "crn_copy_tiles_f32": [vp, vp, vp, vp, vp, i64, i32, vp],
    "crn_copy_mats_f32": [vp, vp, vp, i64, i32, vp],
    "crn_gather_f32": [vp, vp, vp, i64, vp],
    "crn_scatter_f32": [vp, vp, vp, i64, i32, vp],
    "crn_bias_grad": [vp, i32, i32, i64, i64, vp, i32, vp, sz, vp],
    "crn_batch_renorm_stats": [vp, i32, i32, i64, i64, i32, vp, vp, vp, vp, vp, f32, f32, i32,
                               vp, vp, vp, vp, sz, vp],
    "crn_batch_renorm_stats_tail": [vp, i32, i32, i64, i64, vp, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, vp, sz,
                                    vp, vp, vp, i64, vp, i64, vp, i64, i32, vp, i32, vp],
    "crn_batch_renorm_bwd_head": [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, i32, i64, vp, vp, vp, vp, vp, i64,
                                  vp, vp, i32, vp, i32, vp, sz, vp, i32, vp],
    "crn_batch_renorm_bwd": [vp, i64, vp, i64, i32, i32, i64, i32, i32, vp, vp, vp, vp, vp, i64,
                             vp, vp, i32, vp, i32, vp, sz, vp],
    "crn_affine_add_relu": [vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, vp, i64, vp, i64, i32, vp],
    "crn_relu_bwd_add": [vp, vp, vp, i32, i32, i64, i64, i64, i64, vp, i64, vp],
    "crn_preprocess_caffe": [vp, i32, i32, i32, vp, vp],
    "crn_stem_conv_fwd": [vp, i32, i32, i32, vp, vp, vp, vp, sz, vp],
    "crn_stem_conv_wgrad": [vp, i32, i32, i32, vp, vp, vp],
    "crn_batch_renorm_finalize": [vp, i32, i32, C.c_double, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp],
    "crn_bn_relu_maxpool_fwd": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp],
    "crn_bn_relu_maxpool_bwd": [vp, vp, i32, i32, i32, i32, vp, vp],
    "crn_relu_mean_fwd": [vp, i32, i32, i64, i64, vp, vp],
    "crn_relu_mean_bwd": [vp, vp, i32, i32, i64, i64, vp, i64, i32, vp],
    "crn_linear_fwd": [vp, vp, vp, i32, i32, i32, vp, i32, vp],
    "crn_linear_bwd": [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp],
    "crn_stride2_gather": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "crn_stride2_scatter": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "crn_fill_offset_channels": [vp, i32, i64, i64, i32, vp, vp],
    "crn_decoder_inputs": [vp, vp, i32, i32, C.POINTER(C.c_float), vp, vp, vp],
    "crn_ray_sample_fwd": [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, i64, i32, i32, i32, vp],
    "crn_ray_sample_bwd": [vp, i64, i32, i32, i32, i32, i32, vp, vp, vp, i64, i32, i32, i32, vp],
    "crn_ray_sample_fwd_idx": [vp, i64, i64, i64, i32, i32, i32, i32, vp, vp, vp, i64, i32, i32, i32, vp, vp],
    "crn_ray_project": [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_ray_sample_bwd_idx": [vp, i64, i32, i32, i32, i32, i32, vp, vp, i64, i32, i32, i32, vp],
    "crn_loss_fwd_bwd": [i32, vp, vp, vp, i32, i32, i64, vp, vp, f32, vp, sz, vp],
    "crn_argmax_confusion": [vp, vp, i32, i32, i64, vp, vp, vp],
    "crn_softmax_superres": [vp, i32, i32, i32, i32, i32, i32, vp, vp],
    "crn_adam_step": [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp],
    "crn_adam_set_hyper": [vp, f32, f32, f32, f32, f32, i32, vp],
    "crn_adam_step_hyper": [vp, vp, vp, vp, i64, vp, vp],
    "crn_fill_voxels": [vp, vp, i32, i32, i32, i32, i32, vp, sz, vp],
    "crn_fill_voxels_strided": [vp, C.POINTER(i64), vp, C.POINTER(i64), i32, i32, i32, i32, i32, vp, sz, vp],
    "crn_fill_voxels_cpu": [vp, vp, i32, i32, i32, i32, i32, i32],
    "crn_voxelize_mesh": [vp, vp, i32, vp, i32, i32, i32, i32, i32, f32, i32, i32, vp, vp],
    "crn_batch_renorm_eval_affine": [vp, vp, vp, i32, f32, vp, vp, vp],
    "crn_transform_meshes": [vp, vp, i32, vp, i32, vp, vp],
    "crn_merge_labels": [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "crn_zero_f32": [vp, i64, vp],
    "crn_add_i64": [vp, i32, i64, vp],
}
_SIZE_FNS = {
    "crn_batch_renorm_workspace_bytes": [i32],
    "crn_loss_workspace_bytes": [i32, i32],
    "crn_fill_voxels_workspace_bytes": [i32, i32, i32, i32],
    "crn_convt_s2k7_ximage_bytes": [i32, i32, i32, i32],
    "crn_stem_conv_parts": [i32, i32, i32],
}
_PTR_FNS = {
    "crn_loss_status_ptr": [vp, i32],
}
# tuning aids of tools/: only in tools/_build/libcorenet_hip_tools.so (CRN_TOOLS_LIB=1 loads that build instead; reached through
# `.cdll`, no error check: they return CRN_EINVAL unless their switch is set).  The product library does not export them.
TOOL_SYMBOLS = ["crn_bf3_debug_stamps", "crn_e2d_debug_stamps", "crn_pw_debug_stamps"]
ALL_SYMBOLS = list(_SIGS) + list(_SIZE_FNS) + list(_PTR_FNS) + ["crn_version"]


class _Lib:
  def __init__(self, path: str):
    if not os.path.exists(path):
      raise ImportError(
          f"{path} not found: build it with `python -m corenet_amd.build` "
          "(corenet_amd has no CPU or eager-PyTorch fallback).")
    self.path = path
    self.cdll = C.CDLL(path)
    for name, sig in _SIGS.items():
      fn = getattr(self.cdll, name)
      fn.argtypes = sig
      fn.restype = C.c_int
      setattr(self, "_" + name, fn)
    for name, sig in _SIZE_FNS.items():
      fn = getattr(self.cdll, name)
      fn.argtypes = sig
      fn.restype = C.c_size_t
      setattr(self, name, fn)
    for name, sig in _PTR_FNS.items():
      fn = getattr(self.cdll, name)
      fn.argtypes = sig
      fn.restype = C.c_void_p
      setattr(self, name, fn)
    self.cdll.crn_version.restype = C.c_char_p

  def version(self) -> str:
    return self.cdll.crn_version().decode()

  def __getattr__(self, name):
    # crn_xxx(...) with status checking
    if name.startswith("crn_"):
      fn = object.__getattribute__(self, "_" + name)

      def call(*args):
        rc = fn(*args)
        if rc != 0:
          raise HipError(f"{name} failed with status {rc}")
      setattr(self, name, call)
      return call
    raise AttributeError(name)


_lib: Optional[_Lib] = None


def lib() -> _Lib:
  global _lib
  if _lib is None:
    _lib = _Lib(LIB_PATH)
  return _lib


def ptr(x: Optional[t.Tensor]) -> Optional[int]:
  """Device pointer of a tensor (None -> NULL)."""
  if x is None:
    return None
  return x.data_ptr()


import threading

_tls = threading.local()       # the pinned handle is per host thread: autograd runs backward on a worker thread, a
                               # data-loader thread may call the voxelizer on another stream / device at the same time


def stream() -> int:
  """The current torch HIP stream as a hipStream_t value.  Inside pinned_stream() blocks (of THIS thread) it is the
  pinned handle: torch.cuda.current_stream() costs ~1.2 us per call, 0.7 ms of the host's 6 ms per training step."""
  s = getattr(_tls, "stream", None)
  if s is not None:
    return s
  return t.cuda.current_stream().cuda_stream


class pinned_stream:
  """with pinned_stream(): every library call this thread makes inside the block goes to the torch stream that is
  current at entry (or to `s`), without asking torch each time.  Blocks nest; code that switches torch streams inside
  must enter a new block (Plan does, around its side-stream sections).  Other threads are not affected: they keep
  resolving torch.cuda.current_stream() (or their own pin)."""

  def __init__(self, s: Optional[t.cuda.Stream] = None):
    self.s = s

  def __enter__(self):
    self.prev = getattr(_tls, "stream", None)
    _tls.stream = (self.s if self.s is not None else t.cuda.current_stream()).cuda_stream
    return self

  def __exit__(self, *exc):
    _tls.stream = self.prev
    return False


ROCTX = os.environ.get("CRN_ROCTX", "0") == "1"


class roctx_range:
  """with roctx_range("fwd encoder.stage2.a.op_a.conv."): a rocprofv3 marker range around the library calls of the
  block (crn_roctx_push / crn_roctx_pop); does nothing unless CRN_ROCTX=1 (or force) and a roctx library is present."""
  __slots__ = ("label", "on")

  def __init__(self, label: str, force: bool = False):
    self.label, self.on = label, (ROCTX or force)

  def __enter__(self):
    if self.on:
      try:
        lib().crn_roctx_push(self.label.encode())
      except HipError:
        self.on = False
    return self

  def __exit__(self, *exc):
    if self.on:
      try:
        lib().crn_roctx_pop()
      except HipError:
        pass
    return False


def require_gpu(x: t.Tensor, what: str = "tensor"):
  if not x.is_cuda:
    raise ValueError(f"{what}: Only CUDA(HIP) tensors are supported by the corenet_amd kernels")
