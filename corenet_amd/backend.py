"""HipBackend: typed Python front of the C ABI (one method per entry point).

All tensors are torch CUDA(HIP) tensors used as device memory; every call goes to
libcorenet_hip.so on the current torch stream.  No torch compute ops here.
tests/kernel_contract_emu.py implements the same interface with torch-CPU ops as
an executable specification of each kernel's contract (test-only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch as t

from corenet_amd import _lib
from corenet_amd._lib import CrnInTransform, CrnView, ptr
from corenet_amd.views import View

_DTYPE_CODE = {t.float32: 0, t.uint8: 1, t.int32: 2, t.float64: 3, t.int64: 4, t.int16: 5, t.int8: 6}


def _cview(v: View) -> CrnView:
  """The crnView struct of a View, built once per View object and storage address (a step makes ~900 of them;
  building one costs ~3 us of the host's ~11 us per launch)."""
  base = v.storage.data_ptr() + 4 * (v.offset - v.storage.storage_offset())
  c = v.__dict__.get("_crn")
  if c is None or c[0] != base:
    c = (base, CrnView(base, v.B, v.C, v.D, v.H, v.W, v.sB, v.sC, ptr(v.chan_off), v.sD, v.sH, v.sW))
    object.__setattr__(v, "_crn", c)
  return c[1]


class Transform:
  """Host mirror of crnInTransform."""
  __slots__ = ("scale", "shift", "pre_relu", "post_relu")

  def __init__(self, scale: t.Tensor, shift: t.Tensor, pre_relu=False, post_relu=False):
    self.scale, self.shift = scale, shift
    self.pre_relu, self.post_relu = bool(pre_relu), bool(post_relu)


def _ctr(tr: Optional[Transform]):
  if tr is None:
    return None
  return C.byref(CrnInTransform(ptr(tr.scale), ptr(tr.shift), int(tr.pre_relu), int(tr.post_relu)))


_TAPBOXES = {}


def _ctapboxes(boxes):
  """(n_boxes, c_boxes) -> cached crnTapBoxes pointer (None -> NULL)."""
  if not boxes or not (boxes[0] or boxes[1]):
    return None
  key = (tuple(boxes[0]), tuple(boxes[1]))
  tb = _TAPBOXES.get(key)
  if tb is None:
    tb = _lib.CrnTapBoxes()
    tb.n_groups, tb.c_groups = len(key[0]), len(key[1])
    for g, bx in enumerate(key[0]):
      for i, v in enumerate(bx): tb.n_box[g][i] = v
    for g, bx in enumerate(key[1]):
      for i, v in enumerate(bx): tb.c_box[g][i] = v
    _TAPBOXES[key] = tb
  return C.cast(C.pointer(tb), C.c_void_p)


class HipBackend:
  name = "hip"

  def __init__(self):
    self.lib = _lib.lib()
    self._ws = {}
    self._deterministic = False

  # -- workspaces -----------------------------------------------------------
  def workspace(self, key: str, nbytes: int, device) -> t.Tensor:
    w = self._ws.get((key, str(device)))
    if w is None or w.numel() < nbytes:
      w = t.empty(max(nbytes, 1), dtype=t.uint8, device=device)
      self._ws[(key, str(device))] = w
    return w

  def _bn_ws(self, Cn: int, device):
    n = self.lib.crn_batch_renorm_workspace_bytes(Cn)
    # one workspace per stream: the engine runs the skip path's bias gradients on its side stream next to the
    # BatchRenorm backward of the main chain
    return self.workspace("bn@%x" % _lib.stream(), n, device), n

  # -- convolution engine -----------------------------------------------------
  def conv_fwd(self, x: View, tr: Optional[Transform], w: t.Tensor, npad: int,
               bias: Optional[t.Tensor], bias_sB: int, y: View, window, pad_lo,
               splits: int = 1, accumulate: bool = False, boxes=None, math: str = "fp32", wslab=None):
    """boxes: (n_boxes, c_boxes) tuples of conv_geometry.Geom (structural zeros of transposed convs).
    math: "fp32" (v_mfma_f32_16x16x4_f32, the parity default) or "bf16x3" (split-bf16 MFMA engine; wslab: the
    weights pre-arranged by bf3_operands in slab order -- same results, cheaper staging)."""
    if math == "bf16x3" and wslab is not None:
      self.lib.crn_conv_fwd_bf3_slabs(C.byref(_cview(x)), _ctr(tr), ptr(wslab), npad, ptr(bias), bias_sB,
                                      C.byref(_cview(y)), window[0], window[1], window[2],
                                      pad_lo[0], pad_lo[1], pad_lo[2], int(accumulate), _ctapboxes(boxes), _lib.stream())
      return
    if math == "bf16x3":
      self.lib.crn_conv_fwd_bf3(C.byref(_cview(x)), _ctr(tr), ptr(w), npad, ptr(bias), bias_sB,
                                C.byref(_cview(y)), window[0], window[1], window[2],
                                pad_lo[0], pad_lo[1], pad_lo[2], int(accumulate), _ctapboxes(boxes), _lib.stream())
      return
    self.lib.crn_conv_fwd(C.byref(_cview(x)), _ctr(tr), ptr(w), npad, ptr(bias), bias_sB,
                          C.byref(_cview(y)), window[0], window[1], window[2],
                          pad_lo[0], pad_lo[1], pad_lo[2], splits, int(accumulate), _ctapboxes(boxes),
                          _lib.stream())

  def conv_fwd_stats(self, x: View, tr: Optional[Transform], wslab: t.Tensor, npad: int, bias, bias_sB: int, y: View, window,
                     pad_lo, boxes, Cn: int, pre_relu: bool) -> int:
    """conv_fwd(math="bf16x3", wslab) that also leaves the partial sums of the BatchRenorm behind it in this stream's
    BatchRenorm workspace (crn_conv_fwd_bf3_slabs_stats) -> number of parts for bn_finalize, 0: run bn_stats."""
    ws, n = self._bn_ws(Cn, y.storage.device)
    parts = C.c_int(0)
    self.lib.crn_conv_fwd_bf3_slabs_stats(C.byref(_cview(x)), _ctr(tr), ptr(wslab), npad, ptr(bias), bias_sB,
                                          C.byref(_cview(y)), window[0], window[1], window[2], pad_lo[0], pad_lo[1], pad_lo[2],
                                          0, _ctapboxes(boxes), int(pre_relu), ptr(ws), n, C.byref(parts), _lib.stream())
    return parts.value

  def conv_dgrad_bn_bwd(self, dyv: View, wslab: t.Tensor, npad: int, gv: View, g: t.Tensor, window, pad_lo, boxes,
                        x, sB_x, B, Cn, S, pre_relu, gamma, scale, shift, saved, dx, sB_dx, dgamma, dbeta,
                        dsum=None, ndsum=0) -> bool:
    """Data gradient g = conv(dyv) on the split-bf16 engine (gv = the dense view of g) together with the first pass of the
    backward of the BatchRenorm whose output gradient g is (x = the norm's input, dense [B][Cn][S]; decoder blocks,
    reconstruction_decoder.py:56-60): the conv launch leaves the norm's two reduction sums per workgroup
    (crn_conv_fwd_bf3_slabs_bnbwd) and the norm runs its second pass only (crn_batch_renorm_bwd_apply).  Returns False
    when the conv launch could not produce the sums (g is written all the same): the caller runs bn_bwd."""
    ws, n = self._bn_ws(Cn, x.device)
    f = _lib.CrnBnBwdFuse(ptr(x), sB_x, ptr(saved), int(pre_relu), ptr(ws), n, ptr(dsum), int(ndsum), 0)
    self.lib.crn_conv_fwd_bf3_slabs_bnbwd(C.byref(_cview(dyv)), None, ptr(wslab), npad, None, 0, C.byref(_cview(gv)),
                                          window[0], window[1], window[2], pad_lo[0], pad_lo[1], pad_lo[2], 0,
                                          _ctapboxes(boxes), C.byref(f), _lib.stream())
    if f.nparts <= 0:
      return False
    self.lib.crn_batch_renorm_bwd_apply(ptr(x), sB_x, ptr(g), Cn * S, B, Cn, S, int(pre_relu), ptr(gamma),
                                        ptr(scale), ptr(shift), ptr(saved), ptr(dx), sB_dx, ptr(dgamma), ptr(dbeta),
                                        0, ptr(dsum), int(ndsum), ptr(ws), n, f.nparts, _lib.stream())
    return True

  def splitk_defer(self, on: bool = True):
    """The next conv call may leave its split-K sum to the BatchRenorm launch that follows (crn_splitk_defer)."""
    self.lib.crn_splitk_defer(int(on))

  def set_deterministic(self, on: bool = True):
    """Order-independent sums everywhere (crn_set_deterministic): two runs from the same state are bit-identical."""
    self.lib.crn_set_deterministic(int(on))
    self._deterministic = bool(on)

  def splitk_reserve(self, stream: t.cuda.Stream, floats: int = 0):
    """Give `stream` its split-K scratch ahead of a HIP-graph capture (crn_splitk_reserve)."""
    self.lib.crn_splitk_reserve(int(floats), stream.cuda_stream)

  def splitk_release(self, stream: t.cuda.Stream):
    """Give the split-K scratch of `stream` back (crn_splitk_release) and forget its BatchRenorm workspace."""
    self.lib.crn_splitk_release(stream.cuda_stream)
    for k in [k for k in self._ws if k[0] == "bn@%x" % stream.cuda_stream]:
      del self._ws[k]

  def bf3_operands(self, packed: t.Tensor, table, out: t.Tensor):
    """table = (desc int64 [n, 6] on the device, total workgroups): conv_geometry.operand_table."""
    desc, blocks = table
    self.lib.crn_bf3_operands(ptr(packed), ptr(desc), desc.shape[0], blocks, ptr(out), _lib.stream())

  def bf3_gather_image(self, src: t.Tensor, table: t.Tensor, out: t.Tensor, host_table=None):
    """out (bytes) <- bf16 hi / lo entries of src gathered through `table` (int32 [n, 10] on the device:
    conv_geometry.convt_par_*_table)."""
    self.lib.crn_bf3_gather_image(ptr(src), ptr(table), table.shape[0], ptr(out), _lib.stream())

  def convt_par_fwd(self, x: t.Tensor, tr: Optional[Transform], wimg: t.Tensor, bias: Optional[t.Tensor], y: t.Tensor, cout: int,
                    host_table=None, resident: bool = False):
    """ConvTranspose3d(16 -> cout, k 7, stride 2, padding 3, output_padding 1) of T(x) into channels [0, cout) of y
    (csrc/convt_par.hip).  x [B,16,D,H,W] dense inside a sample, y [B,>=cout,2D,2H,2W].  resident: the two-class kernel whose
    weights stay in LDS (image from conv_geometry.convt_res_*_table instead of convt_par_*_table)."""
    B, cin, D, H, W = x.shape
    assert cin == 16 and x[0].is_contiguous() and y[0].is_contiguous() and tuple(y.shape[2:]) == (2 * D, 2 * H, 2 * W)
    fn = self.lib.crn_convt_s2k7_c2_fwd_bf3 if resident else self.lib.crn_convt_s2k7_fwd_bf3
    fn(ptr(x), x.stride(0), B, D, H, W, _ctr(tr), ptr(wimg), ptr(bias), ptr(y), y.stride(0), y.stride(1), cout, _lib.stream())

  def convt_par_dgrad(self, dy: t.Tensor, cout: int, wimg: t.Tensor, dx: t.Tensor, accumulate: bool = False, host_table=None,
                      resident: bool = False):
    """Data gradient of the same layer: dy [B,>=cout,2D,2H,2W] (channels [0, cout)) -> dx [B,16,D,H,W]."""
    B, cin, D, H, W = dx.shape
    assert cin == 16 and dx[0].is_contiguous() and dy[0].is_contiguous() and tuple(dy.shape[2:]) == (2 * D, 2 * H, 2 * W)
    fn = self.lib.crn_convt_s2k7_c2_dgrad_bf3 if resident else self.lib.crn_convt_s2k7_dgrad_bf3
    fn(ptr(dy), dy.stride(0), dy.stride(1), cout, B, D, H, W, ptr(wimg), ptr(dx), dx.stride(0), int(accumulate), _lib.stream())

  def conv2d_bf3(self, x: View, tr: Optional[Transform], wop: t.Tensor, npad: int, bias: Optional[t.Tensor],
                 bias_sB: int, y: View, window, pad_lo, accumulate: bool = False):
    """The encoder engine (csrc/conv_e2d.hip): 1x1 / 3x3 stride-1 convs on operand blocks from bf3_operands."""
    assert window[0] == 1 and pad_lo[0] == 0
    self.lib.crn_conv2d_bf3(C.byref(_cview(x)), _ctr(tr), ptr(wop), npad, ptr(bias), bias_sB, C.byref(_cview(y)),
                            window[1], window[2], pad_lo[1], pad_lo[2], int(accumulate), _lib.stream())

  def convt_ximage(self, x: t.Tensor, tr: Optional[Transform], img: Optional[t.Tensor] = None) -> Optional[t.Tensor]:
    """Operand image of T(x) for the parity-walk weight gradient (crn_convt_s2k7_ximage): x [B,16,D,H,W] dense inside a sample.
    Returns the image (`img` when it is large enough, else a fresh buffer the caller should keep) or None when the path is off
    (CRN_CT_XIMG=0, deterministic mode)."""
    if os.environ.get("CRN_CT_XIMG", "1") == "0" or self._deterministic:
      return None
    B, cin, D, H, W = x.shape
    assert cin == 16 and x[0].is_contiguous()
    nb = self.lib.crn_convt_s2k7_ximage_bytes(B, D, H, W)
    if img is None or img.numel() < nb:
      img = t.empty(nb, dtype=t.uint8, device=x.device)
    self.lib.crn_convt_s2k7_ximage(ptr(x), x.stride(0), B, D, H, W, _ctr(tr), ptr(img), nb, _lib.stream())
    return img

  def conv_wgrad(self, x: View, tr: Optional[Transform], dy: View, dw: t.Tensor, npad: int,
                 window, pad_lo, zero_first: bool = True, boxes=None, math: str = "fp32", ximg: Optional[t.Tensor] = None):
    if math == "bf16x3_1x1":       # 1x1 layers: both operands straight from HBM (csrc/conv_e2d.hip)
      self.lib.crn_conv_wgrad_1x1_bf3(C.byref(_cview(x)), _ctr(tr), C.byref(_cview(dy)), ptr(dw), npad, int(zero_first),
                                      _lib.stream())
      return
    if math == "bf16x3_2d":        # ... 1x1 and 3x3 layers of the encoder
      self.lib.crn_conv_wgrad_2d_bf3(C.byref(_cview(x)), _ctr(tr), C.byref(_cview(dy)), ptr(dw), npad, window[1],
                                     window[2], pad_lo[1], pad_lo[2], int(zero_first), _lib.stream())
      return
    if math == "ct_par":           # decoder stage_6.t1 with > 8 classes: the parity-walk weight gradient (csrc/convt_par.hip); x: the plain view
      xs, ds = x.storage, dy.storage     # of the layer's input, dy: the space-to-depth view of the output gradient tensor
      xp = xs.data_ptr() + 4 * (x.offset - xs.storage_offset())
      img = ximg                   # (the caller made it earlier, e.g. under the forward pass on its side stream)
      if img is None and os.environ.get("CRN_CT_XIMG", "1") != "0" and not self._deterministic:
        # T(x) transformed and split once (crn_convt_s2k7_ximage) instead of in every workgroup of the four parity pairs
        nb = self.lib.crn_convt_s2k7_ximage_bytes(x.B, x.D, x.H, x.W)
        img = self.workspace("ct_ximg", nb, xs.device)
        self.lib.crn_convt_s2k7_ximage(xp, x.sB, x.B, x.D, x.H, x.W, _ctr(tr), ptr(img), nb, _lib.stream())
      rc = self.lib._crn_convt_s2k7_wgrad_bf3(xp, x.sB, x.B, x.D, x.H, x.W, _ctr(tr),
                                              ds.data_ptr(), ds.stride(0), ds.stride(1), dy.C // 8, ptr(dw), npad, int(zero_first),
                                              ptr(img), _lib.stream())
      if rc == 0:
        return
      if rc != -1:                 # CRN_EINVAL: deterministic mode or a shape it does not cover -> the generic split-bf16 engine
        raise _lib.HipError(f"crn_convt_s2k7_wgrad_bf3 failed with status {rc}")
      math = "bf16x3"
    if math == "bf16x3":
      self.lib.crn_conv_wgrad_bf3_boxes(C.byref(_cview(x)), _ctr(tr), C.byref(_cview(dy)), ptr(dw), npad,
                                        window[0], window[1], window[2], pad_lo[0], pad_lo[1], pad_lo[2],
                                        int(zero_first), _ctapboxes(boxes), _lib.stream())
      return
    if math == "stem":             # the encoder's stem on its own kernel (csrc/stem_conv.hip); x: the 2x2 space-to-depth view of the image
      rc = self.lib._crn_stem_conv_wgrad(x.storage.data_ptr() + 4 * x.offset, x.B, 2 * x.H, 2 * x.W,
                                         dy.storage.data_ptr() + 4 * dy.offset, ptr(dw), _lib.stream())
      if rc == 0:
        return
      if rc != -1:                 # CRN_EINVAL: a shape it does not cover, or deterministic mode -> the generic engine below
        raise _lib.HipError(f"crn_stem_conv_wgrad failed with status {rc}")
    self.lib.crn_conv_wgrad(C.byref(_cview(x)), _ctr(tr), C.byref(_cview(dy)), ptr(dw), npad,
                            window[0], window[1], window[2], pad_lo[0], pad_lo[1], pad_lo[2],
                            int(zero_first), _ctapboxes(boxes), _lib.stream())

  def copy_tiles(self, src: t.Tensor, dst: t.Tensor, tiles, reverse: bool = False):
    """tiles: (desc int32 [n,6], mask int64 [n], explicit int32 [m]) from conv_geometry.tile_index, optionally followed
    by the LDS-transposed blocks (int32 [k,16], conv_geometry.mat_index) of the parts that are plain transposes."""
    desc, mask, ex = tiles[:3]
    if desc.shape[0]:
      self.lib.crn_copy_tiles_f32(ptr(src), ptr(dst), ptr(desc), ptr(mask), ptr(ex), desc.shape[0], int(reverse),
                                  _lib.stream())
    if len(tiles) > 3 and tiles[3].shape[0]:
      self.lib.crn_copy_mats_f32(ptr(src), ptr(dst), ptr(tiles[3]), tiles[3].shape[0], int(reverse), _lib.stream())

  def gather(self, src: t.Tensor, idx: t.Tensor, dst: t.Tensor):
    self.lib.crn_gather_f32(ptr(src), ptr(idx), ptr(dst), idx.numel(), _lib.stream())

  def scatter(self, src: t.Tensor, idx: t.Tensor, dst: t.Tensor, accumulate=False):
    self.lib.crn_scatter_f32(ptr(src), ptr(idx), ptr(dst), idx.numel(), int(accumulate), _lib.stream())

  def bias_grad(self, dy: t.Tensor, B, Cn, S, sB, db: t.Tensor, accumulate=False):
    ws, n = self._bn_ws(Cn, dy.device)
    self.lib.crn_bias_grad(ptr(dy), B, Cn, S, sB, ptr(db), int(accumulate), ptr(ws), n, _lib.stream())

  # -- BatchRenorm --------------------------------------------------------------
  def bn_stats(self, x: t.Tensor, B, Cn, S, sB, pre_relu, gamma, beta, rmean, rvar, nbt, eps,
               momentum, training, scale, shift, saved):
    ws, n = self._bn_ws(Cn, x.device)
    self.lib.crn_batch_renorm_stats(ptr(x), B, Cn, S, sB, int(pre_relu), ptr(gamma), ptr(beta),
                                    ptr(rmean), ptr(rvar), ptr(nbt), eps, momentum, int(training),
                                    ptr(scale), ptr(shift), ptr(saved), ptr(ws), n, _lib.stream())

  def stem_conv_fwd(self, img: t.Tensor, w_packed: t.Tensor, bias, y: t.Tensor, stats: bool) -> int:
    """ZeroPad2d(3) + Conv2d(3 -> 64, 7x7, stride 2) on its own kernel (crn_stem_conv_fwd); stats: the partial sums of the
    BatchRenorm that follows come out of the same launch -> the number of parts for bn_finalize (0 without)."""
    B, _, H, W = img.shape
    parts = int(self.lib.crn_stem_conv_parts(B, H, W)) if stats else 0
    ws, n = self._bn_ws(64, img.device) if stats else (None, 0)
    self.lib.crn_stem_conv_fwd(ptr(img), B, H, W, ptr(w_packed), ptr(bias), ptr(y), ptr(ws), n, _lib.stream())
    return parts

  def bn_finalize(self, parts: int, Cn: int, count: float, gamma, beta, rmean, rvar, nbt, eps, momentum, scale, shift, saved):
    """Second half of bn_stats from the partial sums the producing launch left in this stream's BatchRenorm workspace."""
    ws, _ = self._bn_ws(Cn, scale.device)
    self.lib.crn_batch_renorm_finalize(ptr(ws), parts, Cn, float(count), ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar),
                                       ptr(nbt), eps, momentum, ptr(scale), ptr(shift), ptr(saved), _lib.stream())

  def bn_stats_tail(self, x, B, Cn, S, sB, gamma, beta, rmean, rvar, nbt, eps, momentum, training, scale, shift, saved,
                    r, rscale, rshift, sB_r, y_pre, sB_pre, y, sB_y, relu, y2=None, W=0):
    """bn_stats(x) + affine_add_relu(x, scale, shift, r, rscale, rshift, ...) in one call (one launch where a workgroup
    owns a channel in registers: crn_batch_renorm_stats_tail); y2: also the stride-2 compaction of y (rows of width W)."""
    ws, n = self._bn_ws(Cn, x.device)
    self.lib.crn_batch_renorm_stats_tail(ptr(x), B, Cn, S, sB, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ptr(nbt),
                                         eps, momentum, int(training), ptr(scale), ptr(shift), ptr(saved), ptr(ws), n,
                                         ptr(r), ptr(rscale), ptr(rshift), sB_r, ptr(y_pre), sB_pre, ptr(y), sB_y,
                                         int(relu), ptr(y2), int(W), _lib.stream())

  def bn_bwd(self, x, sB_x, dy, sB_dy, B, Cn, S, pre_relu, post_relu, gamma, scale, shift, saved,
             dx, sB_dx, dgamma, dbeta, accumulate=False, dsum=None, ndsum=0):
    ws, n = self._bn_ws(Cn, x.device)
    self.lib.crn_batch_renorm_bwd(ptr(x), sB_x, ptr(dy), sB_dy, B, Cn, S, int(pre_relu),
                                  int(post_relu), ptr(gamma), ptr(scale), ptr(shift), ptr(saved),
                                  ptr(dx), sB_dx, ptr(dgamma), ptr(dbeta), int(accumulate),
                                  ptr(dsum), int(ndsum), ptr(ws), n, _lib.stream())

  def bn_bwd_head(self, x, sB_x, dy, sB_dy, g, sB_g, act, sB_act, g2, sB_g2, B, Cn, S, gamma, scale, shift, saved,
                  dx, sB_dx, dgamma, dbeta, accumulate=False, dsum=None, ndsum=0, g_compact=None, W=0):
    """relu_bwd_add(g, act, g2 -> dy) + bn_bwd(x, dy -> dx) in one call (crn_batch_renorm_bwd_head); g_compact: g in the
    compact form of a stride-2 data gradient (g = stride2_scatter(g_compact), planes with rows of width W)."""
    ws, n = self._bn_ws(Cn, x.device)
    self.lib.crn_batch_renorm_bwd_head(ptr(x), sB_x, ptr(dy), sB_dy, ptr(g), sB_g, ptr(act), sB_act, ptr(g2), sB_g2,
                                       B, Cn, S, ptr(gamma), ptr(scale), ptr(shift), ptr(saved), ptr(dx), sB_dx,
                                       ptr(dgamma), ptr(dbeta), int(accumulate), ptr(dsum), int(ndsum), ptr(ws), n,
                                       ptr(g_compact), int(W), _lib.stream())

  def bn_eval_affine(self, params, buffers, table, eps, scale, shift):
    self.lib.crn_batch_renorm_eval_affine(ptr(params), ptr(buffers), ptr(table), table.shape[0], eps,
                                          ptr(scale), ptr(shift), _lib.stream())

  def affine_add_relu(self, x, scale, shift, r, rscale, rshift, B, Cn, S, sB_x, sB_r,
                      y_pre, sB_pre, y, sB_y, relu):
    self.lib.crn_affine_add_relu(ptr(x), ptr(scale), ptr(shift), ptr(r), ptr(rscale), ptr(rshift),
                                 B, Cn, S, sB_x, sB_r, ptr(y_pre), sB_pre, ptr(y), sB_y, int(relu),
                                 _lib.stream())

  def relu_bwd_add(self, dy, y_pre, dy2, B, Cn, S, sB_dy, sB_pre, sB_dy2, dx, sB_dx):
    self.lib.crn_relu_bwd_add(ptr(dy), ptr(y_pre), ptr(dy2), B, Cn, S, sB_dy, sB_pre, sB_dy2,
                              ptr(dx), sB_dx, _lib.stream())

  # -- encoder odds and ends -------------------------------------------------------
  def preprocess(self, img_u8, out):
    B, _, H, W = img_u8.shape
    self.lib.crn_preprocess_caffe(ptr(img_u8), B, H, W, ptr(out), _lib.stream())

  def maxpool_fwd(self, x, scale, shift, B, Cn, H, W, y, argmax):
    self.lib.crn_bn_relu_maxpool_fwd(ptr(x), ptr(scale), ptr(shift), B, Cn, H, W, ptr(y), ptr(argmax),
                                     _lib.stream())

  def maxpool_bwd(self, dy, argmax, B, Cn, H, W, dx):
    self.lib.crn_bn_relu_maxpool_bwd(ptr(dy), ptr(argmax), B, Cn, H, W, ptr(dx), _lib.stream())

  def relu_mean_fwd(self, x_pre, B, Cn, S, sB, avg):
    self.lib.crn_relu_mean_fwd(ptr(x_pre), B, Cn, S, sB, ptr(avg), _lib.stream())

  def relu_mean_bwd(self, x_pre, davg, B, Cn, S, sB, dx, sB_dx, accumulate=False):
    self.lib.crn_relu_mean_bwd(ptr(x_pre), ptr(davg), B, Cn, S, sB, ptr(dx), sB_dx, int(accumulate),
                               _lib.stream())

  def linear_fwd(self, x, w, bias, B, K, N, y, ldy):
    self.lib.crn_linear_fwd(ptr(x), ptr(w), ptr(bias), B, K, N, ptr(y), ldy, _lib.stream())

  def linear_bwd(self, x, w, dy, lddy, B, K, N, dx, dw, db):
    self.lib.crn_linear_bwd(ptr(x), ptr(w), ptr(dy), lddy, B, K, N, ptr(dx), ptr(dw), ptr(db),
                            _lib.stream())

  def stride2_gather(self, x, y):
    """y[b,c,i,j] = x[b,c,2i,2j] (contiguous tensors; y is ceil(x / 2) in both extents)."""
    B, Cn, h, w = y.shape
    self.lib.crn_stride2_gather(ptr(x), ptr(y), B, Cn, h, w, x.shape[2], x.shape[3], _lib.stream())

  def stride2_scatter(self, dy, dx):
    """dx[b,c,2i,2j] = dy[b,c,i,j], zeros elsewhere."""
    B, Cn, h, w = dy.shape
    self.lib.crn_stride2_scatter(ptr(dy), ptr(dx), B, Cn, h, w, dx.shape[2], dx.shape[3], _lib.stream())

  def decoder_inputs(self, v2s, offset, scales, layer_mats, offset_out):
    """layer_mats[s][b] = v2s[b] . scale(scales[s]) and offset_out = offset in one launch (crn_decoder_inputs)."""
    n = len(scales)
    arr = (C.c_float * n)(*[float(v) for v in scales])
    self.lib.crn_decoder_inputs(ptr(v2s), ptr(offset), v2s.shape[0], n, arr, ptr(layer_mats), ptr(offset_out), _lib.stream())

  def fill_offset_channels(self, x, B, sB, S, c0, offset):
    self.lib.crn_fill_offset_channels(ptr(x), B, sB, S, c0, ptr(offset), _lib.stream())

  # -- ray-traced skip ------------------------------------------------------------------
  def ray_sample_fwd(self, fmap, map_sB, B, Cn, h, w, matrix, offset, out, out_sB, D, H, W,
                     map_sC=None, map_sP=1):
    """map element (b, c, iy, ix) at b*map_sB + c*map_sC + (iy*w+ix)*map_sP; default [B][C][h][w]."""
    self.lib.crn_ray_sample_fwd(ptr(fmap), map_sB, h * w if map_sC is None else map_sC, map_sP, B, Cn, h, w,
                                ptr(matrix), ptr(offset), ptr(out), out_sB, D, H, W, _lib.stream())

  def ray_sample_bwd(self, dout, dout_sB, B, Cn, D, H, W, matrix, offset, dmap, dmap_sB, h, w,
                     zero_first=True):
    self.lib.crn_ray_sample_bwd(ptr(dout), dout_sB, B, Cn, D, H, W, ptr(matrix), ptr(offset),
                                ptr(dmap), dmap_sB, h, w, int(zero_first), _lib.stream())

  def ray_sample_fwd_idx(self, fmap, map_sB, B, Cn, h, w, matrix, offset, out, out_sB, D, H, W, idx,
                         map_sC=None, map_sP=1):
    """ray_sample_fwd that also leaves the saved index tensor idx (uint16 [B][D][H][W]; int16 storage is fine) of the backward."""
    self.lib.crn_ray_sample_fwd_idx(ptr(fmap), map_sB, h * w if map_sC is None else map_sC, map_sP, B, Cn, h, w,
                                    ptr(matrix), ptr(offset), ptr(out), out_sB, D, H, W, ptr(idx), _lib.stream())

  def ray_project(self, matrix, offset, B, D, H, W, h, w, idx):
    self.lib.crn_ray_project(ptr(matrix), ptr(offset), B, D, H, W, h, w, ptr(idx), _lib.stream())

  def ray_sample_bwd_idx(self, dout, dout_sB, B, Cn, D, H, W, idx, dmap, dmap_sB, h, w, zero_first=True):
    self.lib.crn_ray_sample_bwd_idx(ptr(dout), dout_sB, B, Cn, D, H, W, ptr(idx), ptr(dmap), dmap_sB, h, w,
                                    int(zero_first), _lib.stream())

  # -- losses / metrics / optimizer ---------------------------------------------------------
  def loss_fwd_bwd(self, kind, logits, gt_i32, B, Cn, S, loss, dlogits, grad_scale=1.0, weights=None):
    n = self.lib.crn_loss_workspace_bytes(B, Cn)
    ws = self.workspace("loss", n, logits.device)
    self.lib.crn_loss_fwd_bwd(kind, ptr(logits), ptr(gt_i32), ptr(weights), B, Cn, S, ptr(loss), ptr(dlogits),
                              grad_scale, ptr(ws), n, _lib.stream())

  def loss_labels_in_range(self, B, Cn, device) -> bool:
    """True unless the last loss_fwd_bwd on `device` saw a label outside [0, C) (synchronises)."""
    n = self.lib.crn_loss_workspace_bytes(B, Cn)
    ws = self.workspace("loss", n, device)
    off = self.lib.crn_loss_status_ptr(ptr(ws), B) - ws.data_ptr()
    return int(ws[off:off + 4].view(t.int32)) == 0

  def softmax_superres(self, logits, m, B, Cn, D, H, W, out):
    self.lib.crn_softmax_superres(ptr(logits), m, B, Cn, D, H, W, ptr(out), _lib.stream())

  def argmax_confusion(self, logits, gt_i32, B, Cn, S, labels, cm):
    self.lib.crn_argmax_confusion(ptr(logits), ptr(gt_i32), B, Cn, S, ptr(labels), ptr(cm), _lib.stream())

  def adam_step(self, p, g, m, v, n, lr, b1, b2, eps, grad_scale, step):
    self.lib.crn_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), n, lr, b1, b2, eps, grad_scale, step,
                           _lib.stream())

  def adam_set_hyper(self, hyper, lr, b1, b2, eps, grad_scale, step):
    self.lib.crn_adam_set_hyper(ptr(hyper), lr, b1, b2, eps, grad_scale, step, _lib.stream())

  def adam_step_hyper(self, p, g, m, v, n, hyper):
    self.lib.crn_adam_step_hyper(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(hyper), _lib.stream())

  def add_i64(self, p, n, v):
    self.lib.crn_add_i64(ptr(p), n, v, _lib.stream())

  def zero(self, x: t.Tensor):
    self.lib.crn_zero_f32(ptr(x), x.numel(), _lib.stream())

  # -- ground-truth side ------------------------------------------------------------------------
  def fill_voxels(self, grid: t.Tensor, out: t.Tensor):
    N, D, H, W = grid.shape
    n = self.lib.crn_fill_voxels_workspace_bytes(N, D, H, W)
    ws = self.workspace("fill", n, grid.device)
    if grid.is_contiguous() and out.is_contiguous():
      self.lib.crn_fill_voxels(ptr(grid), ptr(out), _DTYPE_CODE[grid.dtype], N, D, H, W, ptr(ws), n,
                               _lib.stream())
    else:      # strided views on either side (the reference's packed accessors, fill_voxels_gpu.cu:146-163)
      import ctypes as C
      gs = (C.c_int64 * 4)(*grid.stride())
      os_ = (C.c_int64 * 4)(*out.stride())
      self.lib.crn_fill_voxels_strided(ptr(grid), gs, ptr(out), os_, _DTYPE_CODE[grid.dtype], N, D, H, W,
                                       ptr(ws), n, _lib.stream())

  def voxelize_mesh(self, tri, tri_mesh, view2voxel, M, D, H, W, sub_side, mult, conservative,
                    depth_mult, grid):
    self.lib.crn_voxelize_mesh(ptr(tri), ptr(tri_mesh), tri.shape[0], ptr(view2voxel), M, D, H, W,
                               sub_side, float(mult), int(conservative), depth_mult, ptr(grid),
                               _lib.stream())

  def transform_meshes(self, triangles, tri_mesh, mesh_matrix, out):
    self.lib.crn_transform_meshes(ptr(triangles), ptr(tri_mesh), triangles.shape[0], ptr(mesh_matrix),
                                  mesh_matrix.shape[0], ptr(out), _lib.stream())

  def merge_labels(self, meshes_grid, scene_start, labels, B, D, H, W, sub_grid, out):
    self.lib.crn_merge_labels(ptr(meshes_grid), ptr(scene_start), ptr(labels), B, D, H, W,
                              int(sub_grid), ptr(out), _lib.stream())


_default: Optional[HipBackend] = None


def default_backend() -> HipBackend:
  global _default
  if _default is None:
    _default = HipBackend()
  return _default
