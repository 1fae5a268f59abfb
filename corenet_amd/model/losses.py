"""Drop-in for `corenet.model.losses` (losses.py:19-179): same function names and
argument meaning, autograd-compatible, computed by the fused HIP loss kernels
(softmax + IoU / cross-entropy reductions + gradient in two streaming passes).

`weights` (per-voxel loss weights) are not used by the reference's training
configs (pipeline.py:228 passes none); passing them raises NotImplementedError
rather than silently falling back to an eager implementation.
"""
from __future__ import annotations

import torch as t

from corenet_amd.backend import default_backend
from corenet_amd.model.engine import LOSS_KINDS


class _LossFn(t.autograd.Function):
  @staticmethod
  def forward(ctx, logits, gt, kind):
    b, c = logits.shape[:2]
    assert logits.dtype == t.float32                      # losses.py:33,81,130
    assert gt.shape == (b,) + tuple(logits.shape[2:]) and gt.dtype in (t.int64, t.int32)
    if not logits.is_cuda:
      raise ValueError("Only CUDA(HIP) tensors are supported by the corenet_amd losses")
    be = default_backend()
    logits = logits.contiguous()
    gt32 = gt.to(t.int32).contiguous()
    S = logits[0, 0].numel()
    loss = t.empty(1, dtype=t.float32, device=logits.device)
    dl = t.empty_like(logits) if logits.requires_grad or t.is_grad_enabled() else None
    be.loss_fwd_bwd(kind, logits, gt32, b, c, S, loss, dl, 1.0)
    ctx.save_for_backward(dl)
    return loss.reshape(())

  @staticmethod
  def backward(ctx, g):
    (dl,) = ctx.saved_tensors
    return dl * g, None, None


def _call(name, gt_volume, logits, weights):
  if weights is not None:
    raise NotImplementedError("per-voxel loss weights are not part of the MI355X hot path")
  return _LossFn.apply(logits, gt_volume, LOSS_KINDS[name])


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
