"""Logical NCDHW views over flat fp32 device buffers (host mirror of crnView).

A View never owns memory: it names `storage` (any torch tensor) plus an element
offset and strides.  Space-to-depth / depth-to-space layouts are expressed with a
per-logical-channel offset table (`chan_off`), so strided convolutions and
transposed convolutions become stride-1 window correlations for the kernels
without materialising the rearranged tensor.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence, Tuple

import numpy as np
import torch as t


@dataclasses.dataclass
class View:
  storage: t.Tensor                 # the tensor whose memory is addressed
  offset: int                       # element offset of (0,0,0,0,0)
  B: int
  C: int
  D: int
  H: int
  W: int
  sB: int
  sC: int
  sD: int
  sH: int
  sW: int
  chan_off: Optional[t.Tensor] = None   # int32 [C] on the storage's device

  @property
  def dims(self) -> Tuple[int, int, int]:
    return self.D, self.H, self.W

  @property
  def spatial(self) -> int:
    return self.D * self.H * self.W

  def channels(self, c0: int, c1: int) -> "View":
    """Sub-range of channels (plain views only)."""
    assert self.chan_off is None
    return dataclasses.replace(self, offset=self.offset + c0 * self.sC, C=c1 - c0)


def view_of(x: t.Tensor) -> View:
  """Plain view of a contiguous-in-space [B,C,H,W] or [B,C,D,H,W] tensor
  (batch stride may be larger than C*S: channel slices of concat buffers)."""
  if x.dim() == 4:
    B, C, H, W = x.shape
    D = 1
    sB, sC, sH, sW = x.stride()
    sD = sC
  else:
    B, C, D, H, W = x.shape
    sB, sC, sD, sH, sW = x.stride()
  return View(x, x.storage_offset(), B, C, D, H, W, sB, sC, sD, sH, sW)


def strided_view(v: View, step: Tuple[int, int, int]) -> View:
  """Every step-th position along D/H/W (1x1 stride-2 convs, resnet50.py:94-97)."""
  sd, sh, sw = step
  return dataclasses.replace(
      v, D=(v.D + sd - 1) // sd, H=(v.H + sh - 1) // sh, W=(v.W + sw - 1) // sw,
      sD=v.sD * sd, sH=v.sH * sh, sW=v.sW * sw)


_TABLES = {}


def _table(vals: np.ndarray, like: t.Tensor) -> t.Tensor:
  """Device copy of a channel-offset table, cached: the same few tables are needed every step."""
  assert vals.max(initial=0) < 2 ** 31
  v = np.ascontiguousarray(vals.astype(np.int32))
  key = (str(like.device), v.shape[0], hash(v.tobytes()))
  tab = _TABLES.get(key)
  if tab is None:
    tab = t.as_tensor(v, device=like.device)
    _TABLES[key] = tab
  return tab


def space_to_depth_view(v: View, r: Tuple[int, int, int], parity_major: bool = False) -> View:
  """Logical channels (c, rd, rh, rw) of a plain view -- or (rd, rh, rw, c) with parity_major, which
  keeps the channels of one sub-position (parity) contiguous -- ; logical spatial dims ceil(dim / r).
  Element (c,rd,rh,rw | q) = v(c | r*q + (rd,rh,rw))."""
  assert v.chan_off is None
  rd, rh, rw = r
  c = np.arange(v.C)[:, None, None, None] * v.sC
  off = (c + np.arange(rd)[None, :, None, None] * v.sD + np.arange(rh)[None, None, :, None] * v.sH
         + np.arange(rw)[None, None, None, :] * v.sW)
  off = (off.transpose(1, 2, 3, 0) if parity_major else off).reshape(-1)
  return dataclasses.replace(
      v, C=v.C * rd * rh * rw, D=(v.D + rd - 1) // rd, H=(v.H + rh - 1) // rh,
      W=(v.W + rw - 1) // rw, sD=v.sD * rd, sH=v.sH * rh, sW=v.sW * rw,
      chan_off=_table(off, v.storage))


def flat_channel_view(v: View) -> View:
  """All of (C,D,H,W) of a plain contiguous view as logical channels of a 1^3 grid."""
  assert v.chan_off is None and v.sW == 1 and v.sH == v.W and v.sC == v.D * v.H * v.W
  n = v.C * v.D * v.H * v.W
  return dataclasses.replace(v, C=n, D=1, H=1, W=1, sC=1, sD=1, sH=1, sW=1)
