"""Drop-in for `corenet.model.losses` (losses.py:19-179): same function names and
argument meaning, autograd-compatible, computed by the fused HIP loss kernels
(softmax + IoU / cross-entropy reductions + gradient in two streaming passes).

`weights` (per-voxel loss weights, float32[B,D,H,W]; losses.py:47-49,99-102,134-136) are
applied inside the same kernels.  Labels outside [0, C) make the reference raise inside
F.one_hot / F.cross_entropy; checking that needs a device->host read-back, so it is opt-in:
with CRN_CHECK_LABELS=1 the loss functions raise ValueError, otherwise such a label is
computed as class 0 (never an out-of-bounds read).
"""
from __future__ import annotations

import os

import torch as t

from corenet_amd.backend import default_backend
from corenet_amd.model.engine import LOSS_KINDS


class _LossFn(t.autograd.Function):
  @staticmethod
  def forward(ctx, logits, gt, kind, weights):
    b, c = logits.shape[:2]
    assert logits.dtype == t.float32                      # losses.py:33,81,130
    assert gt.shape == (b,) + tuple(logits.shape[2:]) and gt.dtype in (t.int64, t.int32)
    if not logits.is_cuda:
      raise ValueError("Only CUDA(HIP) tensors are supported by the corenet_amd losses")
    if weights is not None:                               # losses.py:48,100,135
      assert weights.shape == gt.shape and weights.dtype == t.float32
      weights = weights.to(logits.device).contiguous()
    be = default_backend()
    logits = logits.contiguous()
    gt32 = gt.to(t.int32).contiguous()
    S = logits[0, 0].numel()
    loss = t.empty(1, dtype=t.float32, device=logits.device)
    dl = t.empty_like(logits) if logits.requires_grad or t.is_grad_enabled() else None
    be.loss_fwd_bwd(kind, logits, gt32, b, c, S, loss, dl, 1.0, weights=weights)
    if os.environ.get("CRN_CHECK_LABELS") == "1" and not be.loss_labels_in_range(b, c, logits.device):
      raise ValueError(f"gt_volume holds labels outside [0, {c})")
    ctx.save_for_backward(dl)
    return loss.reshape(())

  @staticmethod
  def backward(ctx, g):
    (dl,) = ctx.saved_tensors
    return dl * g, None, None, None


def _call(name, gt_volume, logits, weights):
  return _LossFn.apply(logits, gt_volume, LOSS_KINDS[name], weights)


def iou_agnostic(gt_volume, logits, weights=None):
  """losses.py:19-61."""
  return _call("iou_agnostic", gt_volume, logits, weights)


def iou_fgbg(gt_volume, logits, weights=None):
  """losses.py:64-114."""
  return _call("iou_fgbg", gt_volume, logits, weights)


def xent(gt_volume, logits, weights=None):
  """losses.py:117-141."""
  return _call("xent", gt_volume, logits, weights)


def xent_times_iou_agnostic(gt_volume, predicted_grid_logits, weights=None):
  """losses.py:144-160."""
  return _call("xent_times_iou_agnostic", gt_volume, predicted_grid_logits, weights)


def xent_times_iou_fgbg(gt_volume, predicted_grid_logits, weights=None):
  """losses.py:163-179."""
  return _call("xent_times_iou_fgbg", gt_volume, predicted_grid_logits, weights)
