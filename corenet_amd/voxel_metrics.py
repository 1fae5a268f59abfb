"""Eval epilogue: fused argmax + confusion matrix on the GPU
(evaluation_results.py:40-51, voxel_metrics.py:33-58,123-138)."""
from __future__ import annotations

import math

import torch as t

from corenet_amd.backend import default_backend


def argmax_confusion(logits: t.Tensor, gt: t.Tensor, num_classes: int, cm: t.Tensor = None):
  """Returns (pred_labels int32[B,...], confusion_matrix int64[K,K]); cm[gt, pred]."""
  B, C = logits.shape[:2]
  S = logits[0, 0].numel()
  assert C == num_classes
  labels = t.empty((B,) + tuple(logits.shape[2:]), dtype=t.int32, device=logits.device)
  if cm is None:
    cm = t.zeros(num_classes, num_classes, dtype=t.int64, device=logits.device)
  default_backend().argmax_confusion(logits.contiguous(), gt.to(t.int32).contiguous(), B, C, S, labels, cm)
  return labels, cm


def mean_iou(cm: t.Tensor, void_class: int = 0) -> float:
  """voxel_metrics.py:123-138 + evaluation_results.py:262-266 (NaN for absent classes,
  mean over non-void classes skipping NaNs like pandas)."""
  cm = cm.to(t.float64).cpu()
  tp = cm.diag(); fp = cm.sum(0) - tp; fn = cm.sum(1) - tp
  vals = [float(tp[i] / (tp[i] + fp[i] + fn[i])) for i in range(cm.shape[0])
          if i != void_class and tp[i] != 0]
  return sum(vals) / len(vals) if vals else math.nan
