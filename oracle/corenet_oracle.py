"""CPU ORACLE for the CoReNet forward/backward hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, with plain torch-CPU / numpy fp32 ops, the arithmetic of
the reference (google-research/corenet, `/root/reference/src/corenet/...`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import it.  The product (`corenet_amd/`) never imports it and has no CPU
fallback.

Pinning status:
  * losses, fill_voxels, voxel_metrics, transformations: pinned by the
    reference's own known-answer tests (tests/golden/reference_known_answers.py)
  * model forward/backward, BatchRenorm, SampleGrid2d: pinned by golden vectors
    generated from the imported reference (oracle/gen_golden.py ->
    tests/golden/*.npz) -- the reference's tests do not cover them.
  * surface voxelizer: pinned ONLY by the three known-answer tests of
    voxelization_test.py:53-147; beyond them "parity unpinned" (the reference
    uses the NVIDIA GL rasterizer, which cannot run here).

Every function cites the reference file:line it follows.  All tensors are
fp32 NCHW / NCDHW like the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch as t
import torch.nn.functional as F

Tensor = t.Tensor
State = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# geometry/transformations.py
# ----------------------------------------------------------------------------
def scale(v: Sequence[float]) -> Tensor:
  """transformations.py:27-39."""
  v = t.as_tensor(v, dtype=t.float32)
  return t.diag(t.cat([v, v.new_ones([1])], 0))


def translate(v) -> Tensor:
  """transformations.py:42-60 (batched)."""
  v = t.as_tensor(v, dtype=t.float32)
  n = v.shape[-1]
  m = t.eye(n + 1, dtype=t.float32).expand(*v.shape[:-1], n + 1, n + 1).clone()
  m[..., :n, n] = v
  return m


def look_at_rh(eye, center, up) -> Tensor:
  """transformations.py:201-220."""
  eye, center, up = [t.as_tensor(x, dtype=t.float32) for x in (eye, center, up)]
  f = F.normalize(center - eye, dim=-1)
  s = F.normalize(t.linalg.cross(f, up), dim=-1)
  u = t.linalg.cross(s, f)
  return t.tensor([
      [s[0], s[1], s[2], -t.dot(s, eye)],
      [u[0], u[1], u[2], -t.dot(u, eye)],
      [-f[0], -f[1], -f[2], t.dot(f, eye)],
      [0, 0, 0, 1]], dtype=t.float32)


def perspective_rh(fov_y, aspect, z_near, z_far) -> Tensor:
  """transformations.py:244-262."""
  fov_y = t.as_tensor(fov_y, dtype=t.float32)
  th = t.tan(fov_y / 2)
  z_near = t.as_tensor(z_near, dtype=t.float32)
  z_far = t.as_tensor(z_far, dtype=t.float32)
  return t.tensor([
      [1.0 / (aspect * th), 0, 0, 0],
      [0, 1.0 / th, 0, 0],
      [0, 0, -(z_far + z_near) / (z_far - z_near),
       -(2 * z_far * z_near) / (z_far - z_near)],
      [0, 0, -1, 0]], dtype=t.float32)


def ortho_lh(left, right, bottom, top, z_near, z_far) -> Tensor:
  """transformations.py:265-286."""
  l, r, b, tp, n, f = [t.as_tensor(x, dtype=t.float32)
                       for x in (left, right, bottom, top, z_near, z_far)]
  return t.tensor([
      [2 / (r - l), 0, 0, -(r + l) / (r - l)],
      [0, 2 / (tp - b), 0, -(tp + b) / (tp - b)],
      [0, 0, 2 / (f - n), -(f + n) / (f - n)],
      [0, 0, 0, 1]], dtype=t.float32)


def canonical_camera() -> Tensor:
  """The dataset's canonical camera, doc/data_format_and_coordinate_systems.md:103-111."""
  return perspective_rh(math.radians(60.0), 1.0, 1e-4, 10.0) @ look_at_rh(
      [0.5, 0.5, -0.8666666], [0.5, 0.5, 0.5], [0, -1, 0])


def transform_points_homogeneous(points: Tensor, matrix: Tensor, w: float) -> Tensor:
  """transformations.py:108-136.  points [B,V,3], matrix [B,4,4] -> [B,V,4]."""
  points = F.pad(points, [0, 1], value=w)
  return t.einsum("bnm,bvm->bvn", matrix, points)


# ----------------------------------------------------------------------------
# model/resnet50.py, model/batch_renorm.py
# ----------------------------------------------------------------------------
def preprocess_image_caffe(image: Tensor) -> Tensor:
  """resnet50.py:189-204.  NB: the means are ADDED."""
  assert image.dtype == t.uint8 and image.dim() == 4 and image.shape[1] == 3
  image = image.to(t.float32).flip(1)
  return image + image.new_tensor([103.939, 116.779, 123.68])[None, :, None, None]


def batch_renorm(x: Tensor, sd: State, prefix: str, training: bool,
                 eps: float = 1e-3, momentum: float = 0.01) -> Tensor:
  """batch_renorm.py:33-62.  Updates sd[prefix+running_*] in place when training."""
  w, b = sd[prefix + "weight"], sd[prefix + "bias"]
  rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
  nbt = sd[prefix + "num_batches_tracked"]
  vd = [1, x.shape[1]] + [1] * (x.dim() - 2)
  _v = lambda v: v.view(vd)
  running_std = (rv + eps).sqrt()
  if training:
    nt = nbt
    d_max = (5.0 * (nt - 5000) / (25000 - 5000)).clamp(0.0, 5.0)
    r_max = 1.0 + (2.0 * (nt - 5000) / (40000 - 5000)).clamp(0.0, 2.0)
    rd = [i for i in range(x.dim()) if i != 1]
    b_mean = x.mean(rd)
    b_var = x.var(rd, unbiased=False)
    b_std = (b_var + eps).sqrt()
    r = (b_std.detach() / running_std).clamp(1 / r_max, r_max)
    d = ((b_mean.detach() - rm) / running_std)
    d = t.max(t.min(d, d_max), -d_max)
    x = (x - _v(b_mean)) / _v(b_std) * _v(r) + _v(d)
    with t.no_grad():
      c = x.shape[1]
      unbiased_var = b_var.detach() * c / (c - 1)   # C = channels
      rv += momentum * (unbiased_var - rv)
      rm += momentum * (b_mean.detach() - rm)
      nbt += 1
  else:
    x = (x - _v(rm)) / _v(running_std)
  return _v(w) * x + _v(b)


def _conv_bn(x, sd, p, training, stride=1, padding=0):
  x = F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=stride,
               padding=padding)
  return batch_renorm(x, sd, p + "bn.", training)


def _identity_block(x, sd, p, training):
  """resnet50.py:49-82."""
  inp = x
  x = _conv_bn(x, sd, p + "op_a.", training).relu()
  x = _conv_bn(x, sd, p + "op_b.", training, padding=1).relu()
  x = _conv_bn(x, sd, p + "op_c.", training) + inp
  return x.relu(), x


def _downscale_block(x, sd, p, training, stride):
  """resnet50.py:85-115."""
  s = _conv_bn(x, sd, p + "shortcut.", training, stride=stride)
  x = _conv_bn(x, sd, p + "op_a.", training, stride=stride).relu()
  x = _conv_bn(x, sd, p + "op_b.", training, padding=1).relu()
  x = _conv_bn(x, sd, p + "op_c.", training) + s
  return x.relu(), x


RESNET_STAGES = (("stage2", "abc", 1), ("stage3", "abcd", 2),
                 ("stage4", "abcdef", 2), ("stage5", "abc", 2))


def resnet50_features(image_f32: Tensor, sd: State, training: bool,
                      prefix: str = "encoder."):
  """resnet50.py:176-186.  Returns (stage2..5 pre-ReLU block outputs, global avg)."""
  p = prefix
  x = F.conv2d(F.pad(image_f32, [3, 3, 3, 3]), sd[p + "stage1.conv.weight"],
               sd[p + "stage1.conv.bias"], stride=2)
  x = batch_renorm(x, sd, p + "stage1_part2.bn.", training).relu()
  x = F.max_pool2d(F.pad(x, [1, 1, 1, 1]), 3, 2)
  feats = []
  for name, blocks, stride in RESNET_STAGES:
    pre = None
    for bl in blocks:
      bp = f"{p}{name}.{bl}."
      if bl == "a":
        x, pre = _downscale_block(x, sd, bp, training, stride)
      else:
        x, pre = _identity_block(x, sd, bp, training)
    feats.append(pre)
  return feats, x.mean(dim=(2, 3))


# ----------------------------------------------------------------------------
# model/ray_traced_skip_connection.py
# ----------------------------------------------------------------------------
def ray_sample_indices(matrix: Tensor, offset: Tensor, grid_dhw, wh):
  """The pixel index each voxel centre gathers from.

  ray_traced_skip_connection.py:91-133.  Arithmetic order FIXED for the HIP
  kernel to reproduce bit-exactly (fp32, no FMA contraction):
      c  = (x, y, z) + offset                          (:96-97)
      p_n = ((m_n0*cx + m_n1*cy) + m_n2*cz) + m_n3     (:103-104, einsum bnm,bvm)
      u  = (p_x / p_w) / 2 + 0.5 ; v likewise          (:109, :112)
      ix = trunc(u * W) ; iy = trunc(v * H)            (:121-122)  
      +1, clamp into the 1-px zero pad                 (:131-132)
      behind-camera test: p_z >= 0                     (:108, :140-142) (Q7)
  Returns int64 (iy_padded, ix_padded) [B,D,H,W] and bool keep [B,D,H,W].
  """
  B = matrix.shape[0]
  D, H, W = grid_dhw
  width, height = wh
  zz, yy, xx = t.meshgrid(t.arange(D, dtype=t.float32), t.arange(H, dtype=t.float32),
                          t.arange(W, dtype=t.float32), indexing="ij")
  m = matrix.to(t.float32)
  o = offset.to(t.float32)
  cx = xx[None] + o[:, 0, None, None, None]
  cy = yy[None] + o[:, 1, None, None, None]
  cz = zz[None] + o[:, 2, None, None, None]
  def row(n):
    mm = lambda j: m[:, n, j, None, None, None]
    return ((mm(0) * cx + mm(1) * cy) + mm(2) * cz) + mm(3)
  px, py, pz, pw = row(0), row(1), row(2), row(3)
  u = (px / pw) / 2 + 0.5
  v = (py / pw) / 2 + 0.5
  ix = (u * float(width)).to(t.int64)
  iy = (v * float(height)).to(t.int64)
  ix = (ix + 1).clamp(0, width + 1)
  iy = (iy + 1).clamp(0, height + 1)
  return iy, ix, pz >= 0


def ray_sample(grid2d: Tensor, matrix: Tensor, offset: Tensor, grid_dhw) -> Tensor:
  """Gather part of SampleGrid2d.forward (ray_traced_skip_connection.py:124-144).

  grid2d: compressed map [B,C,h,w] -> [B,C,D,H,W]."""
  B, C, h, w = grid2d.shape
  iy, ix, keep = ray_sample_indices(matrix, offset, grid_dhw, (w, h))
  padded = F.pad(grid2d, [1, 1, 1, 1])
  bb = t.arange(B)[:, None, None, None].expand_as(iy)
  res = padded[bb, :, iy, ix].permute(0, 4, 1, 2, 3)
  return t.where(keep[:, None], res, t.zeros_like(res))


def sample_grid2d(src2d: Tensor, weight: Tensor, bias: Tensor, matrix: Tensor,
                  offset: Tensor, grid_dhw) -> Tensor:
  """SampleGrid2d.forward: 1x1 compress conv then gather (:85, :124-144)."""
  return ray_sample(F.conv2d(src2d, weight, bias), matrix, offset, grid_dhw)


# ----------------------------------------------------------------------------
# model/reconstruction_decoder.py, model/core_net.py
# ----------------------------------------------------------------------------
def decoder_forward(feats, avg, sd: State, v2s: Tensor, offset: Tensor,
                    resolution, training: bool, prefix="decoder.") -> Tensor:
  """reconstruction_decoder.py:119-152 (+ _apply_skip :97-117)."""
  p = prefix
  f2, f3, f4, f5 = feats
  res = t.tensor(resolution, dtype=t.float32)

  def skip(x, src2d, stage):
    key = f"{p}rt_skip_{stage}.compress_channels."
    if key + "weight" not in sd:
      return x
    o = offset.to(src2d.dtype)[:, :, None, None].expand(src2d.shape[0], 3, *src2d.shape[2:])
    s2 = t.cat([src2d, o], 1)
    r1 = t.tensor(x.shape[2:], dtype=t.float32)
    layer_matrix = v2s.matmul(scale(res / r1))          # Q8
    sk = sample_grid2d(s2, sd[key + "weight"], sd[key + "bias"], layer_matrix,
                       offset, x.shape[2:])
    return t.cat([x, sk], 1)

  def bn(x, name):
    return batch_renorm(x.relu(), sd, f"{p}{name}.", training)

  x = F.linear(avg, sd[p + "stage_0.weight"], sd[p + "stage_0.bias"])
  x = t.cat([x, offset.to(x.dtype)], 1)[:, :, None, None, None]
  x = bn(x, "stage_1.b1")
  ir = resolution[0] // (16 * _last_upscale(sd, p))
  x = F.conv_transpose3d(x, sd[p + "stage_1.t1.weight"], sd[p + "stage_1.t1.bias"],
                         stride=ir)
  x = skip(x, f5, 1)
  pads = {2: 1, 3: 2, 4: 2, 5: 2, 6: 2}
  tpads = {2: 1, 3: 3, 4: 3, 5: 3, 6: 3}
  srcs = {2: f5, 3: f4, 4: f3, 5: f2, 6: None}
  for st in range(2, 7):
    sp = f"{p}stage_{st}."
    x = bn(x, f"stage_{st}.b1")
    x = F.conv3d(x, sd[sp + "c1.weight"], sd[sp + "c1.bias"], padding=pads[st])
    x = bn(x, f"stage_{st}.b2")
    stride = 2 if st < 6 else _last_upscale(sd, p)
    x = F.conv_transpose3d(x, sd[sp + "t1.weight"], sd[sp + "t1.bias"], stride=stride,
                           padding=tpads[st], output_padding=1)
    if srcs[st] is not None:
      x = skip(x, srcs[st], st)
  return x


def _last_upscale(sd, p):
  return 2  # the only value the reference model runs at


def corenet_forward(sd: State, image_u8: Tensor, v2s: Tensor, offset: Tensor,
                    resolution=(128, 128, 128), training: bool = True) -> Tensor:
  """core_net.py:36-43."""
  x = preprocess_image_caffe(image_u8)
  feats, avg = resnet50_features(x, sd, training)
  return decoder_forward(feats, avg, sd, v2s, offset, resolution, training)


# ----------------------------------------------------------------------------
# model/losses.py
# ----------------------------------------------------------------------------
def iou_agnostic(gt: Tensor, logits: Tensor, weights: Optional[Tensor] = None):
  """losses.py:19-61."""
  b, c = logits.shape[:2]
  g = F.one_hot(gt, c).to(t.float32).permute(0, 4, 1, 2, 3)[:, 1:]
  pr = logits.softmax(1)[:, 1:]
  fw = t.where(g == 0, t.ones_like(g), t.ones_like(g) * (c - 1.0))
  if weights is not None:
    fw = fw * weights[:, None]
  inter = (t.min(g, pr) * fw).sum(dim=[1, 2, 3, 4])
  union = (t.max(g, pr) * fw).sum(dim=[1, 2, 3, 4])
  iou = inter / t.where(union == 0, t.ones_like(union), union)
  return 1 - iou.mean()


def iou_fgbg(gt: Tensor, logits: Tensor, weights: Optional[Tensor] = None):
  """losses.py:64-114."""
  b, c = logits.shape[:2]
  g = F.one_hot(gt, c).to(t.float32).permute(0, 4, 1, 2, 3)[:, 1:].sum(1)
  pr = logits.softmax(1)[:, 1:].sum(1)
  g = t.min(g, g.new_tensor(1.0))
  inter, union = t.min(g, pr), t.max(g, pr)
  if weights is not None:
    inter, union = inter * weights, union * weights
  inter = inter.reshape(b, -1).sum(1)
  union = union.reshape(b, -1).sum(1)
  iou = inter / t.where(union == 0, t.ones_like(union), union)
  return 1 - iou.mean()


def xent(gt: Tensor, logits: Tensor, weights: Optional[Tensor] = None):
  """losses.py:117-141."""
  loss = F.cross_entropy(logits, gt, reduction="none")
  if weights is not None:
    loss = loss * weights
  return loss.mean()


def xent_times_iou_agnostic(gt, logits, weights=None):
  """losses.py:144-160."""
  return (1 + iou_agnostic(gt, logits, weights)) * (1 + xent(gt, logits, weights))


def xent_times_iou_fgbg(gt, logits, weights=None):
  """losses.py:163-179."""
  return (1 + iou_fgbg(gt, logits, weights)) * (1 + xent(gt, logits, weights))


# ----------------------------------------------------------------------------
# voxel_metrics.py / evaluation_results.py
# ----------------------------------------------------------------------------
def extract_labels(probs_or_logits: Tensor) -> Tensor:
  """evaluation_results.py:40-51: per-voxel argmax over classes."""
  return probs_or_logits.argmax(dim=1)


def confusion_matrix(gt: Tensor, pred: Tensor, num_classes: int) -> Tensor:
  """voxel_metrics.py:33-58: cm[gt, pred] counts."""
  idx = (gt.reshape(-1).to(t.int64) * num_classes + pred.reshape(-1).to(t.int64))
  return t.bincount(idx, minlength=num_classes * num_classes).reshape(
      num_classes, num_classes)


def mean_iou(cm: Tensor, void_class: int = 0) -> float:
  """voxel_metrics.py:118-138 (`nan_tp_div`: IoU is NaN for a class with tp == 0) +
  evaluation_results.py:262-266 (`mm.iloc[:, 1:-1].T.mean().iou`: pandas mean over the
  non-void classes, NaNs skipped; NaN if every class is NaN)."""
  cm = cm.to(t.float64)
  tp = cm.diag()
  fp = cm.sum(0) - tp
  fn = cm.sum(1) - tp
  iou = t.where(tp == 0, t.full_like(tp, math.nan), tp / (tp + fp + fn))
  keep = t.tensor([i for i in range(cm.shape[0]) if i != void_class], dtype=t.int64)
  vals = iou[keep]
  vals = vals[~vals.isnan()]
  return float(vals.mean()) if vals.numel() else math.nan


# ----------------------------------------------------------------------------
# cc/fill_voxels_{gpu.cu,cpu.cc}: flood fill
# ----------------------------------------------------------------------------
def fill_inside_voxels(grid: np.ndarray) -> np.ndarray:
  """GPU semantics of fill_voxels_gpu.cu:96-132, restated as a BFS.

  A voxel is "outside" iff it is empty (value <= 0) and 6-connected through
  empty voxels to an empty voxel on the x==0, y==0 or z==0 face (the virtual BG
  node is adjacent only to the LOW faces, fill_voxels_gpu.cu:108-119).  Output is
  strictly {0,1}: 0 for outside voxels, 1 for everything else (:131).
  grid: [N,D,H,W] any real dtype.  (oracle/fill_voxels_oracle.c is the fast C twin.)
  """
  from collections import deque
  g = np.asarray(grid)
  N, D, H, W = g.shape
  out = np.ones(g.shape, dtype=g.dtype)
  for n in range(N):
    empty = ~(g[n] > 0)
    seen = np.zeros((D, H, W), bool)
    q = deque()
    def push(z, y, x):
      if empty[z, y, x] and not seen[z, y, x]:
        seen[z, y, x] = True
        q.append((z, y, x))
    for y in range(H):
      for x in range(W):
        push(0, y, x)
    for z in range(D):
      for x in range(W):
        push(z, 0, x)
      for y in range(H):
        push(z, y, 0)
    while q:
      z, y, x = q.popleft()
      if z > 0: push(z - 1, y, x)
      if z + 1 < D: push(z + 1, y, x)
      if y > 0: push(z, y - 1, x)
      if y + 1 < H: push(z, y + 1, x)
      if x > 0: push(z, y, x - 1)
      if x + 1 < W: push(z, y, x + 1)
    out[n][seen] = 0
  return out


# ----------------------------------------------------------------------------
# geometry/voxelization.py + shaders/voxelize.{geom,frag}: surface voxelizer
# ----------------------------------------------------------------------------
def voxelize_mesh(triangles: np.ndarray, mesh_num_tri: Sequence[int], resolution,
                  view2voxel: np.ndarray, sub_grid_sampling=False,
                  image_resolution_multiplier=4, conservative_rasterization=False,
                  projection_depth_multiplier=1, chunk_pixels: int = 1 << 22) -> np.ndarray:
  """Restatement of voxelization.py:98-164 with the rasterizer written out.

  Rules (PARITY UNPINNED beyond voxelization_test.py:53-147 -- the reference
  uses the NVIDIA GL rasterizer):
    * vertices -> voxel space with the mesh's view2voxel (voxelize.geom:31-45)
    * dominant axis of the normal picks the projection plane (voxelize.geom:53-55):
      a.x>a.y && a.x>a.z -> screen=(y,z); a.y>a.x && a.y>a.z -> screen=(z,x);
      else (x,y)
    * ortho_lh(0,W,H,0,0,D*pdm) then viewport of R=round(max(W,H,D*pdm)*mult)
      pixels (voxelization.py:119-120,146-149)
    * a fragment is produced for every pixel whose CENTRE is inside the
      triangle (top-left fill rule); with conservative rasterization for every
      pixel whose square overlaps the triangle (attributes evaluated at the
      pixel centre, extrapolated)
    * fragment stage: bounds test, floor, flat index / sub-grid index
      (voxelize.frag:36-56)

  BIT-DEFINED: every operation is an IEEE fp32 add / sub / mul / div / sqrt in a
  fixed order without FMA contraction (numpy float32 arrays never fuse); the HIP
  rasterizer (csrc/voxelize.hip, built with -ffp-contract=off) performs the same
  operations in the same order, so the two must agree on EVERY voxel, like
  ray_sample_indices.  Vectorised over triangles and over the pixels of their
  bounding boxes (chunked) so that the 128^3 / multiplier 8 / 40 k triangle
  workload finishes in seconds.
  """
  f = np.float32
  tri = np.ascontiguousarray(np.asarray(triangles, f).reshape(-1, 3, 3))
  mnt = np.asarray(mesh_num_tri, np.int64)
  M = len(mnt)
  D, H, W = (int(r) for r in resolution)
  v2v = np.asarray(view2voxel, f)
  if v2v.ndim == 2:
    v2v = np.broadcast_to(v2v, (M, 4, 4))
  if sub_grid_sampling and image_resolution_multiplier % 2 == 0:
    raise ValueError(
        "image_resolution_multiplier must be off if sub_grid_sampling is True")
  shape_index = np.repeat(np.arange(M), mnt)               # misc_util.dynamic_tile
  depth_ext = D * projection_depth_multiplier
  R = int(round(max(W, H, depth_ext) * image_resolution_multiplier))
  vs = int(image_resolution_multiplier) if sub_grid_sampling else -1
  if sub_grid_sampling:
    Dg, Hg, Wg = 2 * D + 1, 2 * H + 1, 2 * W + 1
  else:
    Dg, Hg, Wg = D, H, W
  out = np.zeros((M, Dg, Hg, Wg), np.float32)
  if tri.shape[0] == 0:
    return out
  one, two, half = f(1), f(2), f(0.5)
  Wf, Hf, Df, Def, Rf = f(W), f(H), f(D), f(depth_ext), f(R)

  with np.errstate(all="ignore"):
    A = v2v[shape_index]                                     # [T,4,4]
    x, y, z = tri[:, :, 0], tri[:, :, 1], tri[:, :, 2]       # [T,3 verts]
    # v[t, vert, r] = ((A[r,0]*x + A[r,1]*y) + A[r,2]*z) + A[r,3]
    v = np.stack([((A[:, r, 0, None] * x + A[:, r, 1, None] * y) + A[:, r, 2, None] * z) + A[:, r, 3, None]
                  for r in range(3)], -1)
    e1 = v[:, 1] - v[:, 0]
    e2 = v[:, 2] - v[:, 0]
    n1 = np.sqrt((e1[:, 0] * e1[:, 0] + e1[:, 1] * e1[:, 1]) + e1[:, 2] * e1[:, 2])
    n2 = np.sqrt((e2[:, 0] * e2[:, 0] + e2[:, 1] * e2[:, 1]) + e2[:, 2] * e2[:, 2])
    alive = (n1 != 0) & (n2 != 0)
    e1 = e1 / n1[:, None]
    e2 = e2 / n2[:, None]
    ax_ = np.abs(e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1])
    ay_ = np.abs(e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2])
    az_ = np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    T = tri.shape[0]
    # screen x, screen y, depth <- voxel axes (voxelize.geom:53-55: yzxw / zxyw / xyzw)
    a0 = np.zeros(T, np.int64); a1 = np.ones(T, np.int64); a2 = np.full(T, 2, np.int64)
    cx_ = (ax_ > ay_) & (ax_ > az_)
    cy_ = ~cx_ & (ay_ > ax_) & (ay_ > az_)
    a0[cx_], a1[cx_], a2[cx_] = 1, 2, 0
    a0[cy_], a1[cy_], a2[cy_] = 2, 0, 1
    # transformations.ortho_lh(0, W, H, 0, 0, depth_ext): x: 2x/W-1, y: 1-2y/H, z: 2z/De-1
    ndc = np.stack([two * v[:, :, 0] / Wf - one, one - two * v[:, :, 1] / Hf, two * v[:, :, 2] / Def - one], -1)
    ar = np.arange(T)[:, None]
    sx = (ndc[ar, np.arange(3)[None], a0[:, None]] + one) * half * Rf      # [T,3]
    sy = (ndc[ar, np.arange(3)[None], a1[:, None]] + one) * half * Rf
    sz = ndc[ar, np.arange(3)[None], a2[:, None]]
    area = (sx[:, 1] - sx[:, 0]) * (sy[:, 2] - sy[:, 0]) - (sx[:, 2] - sx[:, 0]) * (sy[:, 1] - sy[:, 0])
    alive &= (area != 0) & ~np.isnan(area)
    sgn = np.where(area > 0, one, -one).astype(f)
    def toint(q):                    # (int) of a finite float; triangles with NaN / inf corners are dropped above
      return np.where(np.isfinite(q), q, 0).astype(np.int64)
    smin = lambda s: np.minimum(s[:, 0], np.minimum(s[:, 1], s[:, 2]))
    smax = lambda s: np.maximum(s[:, 0], np.maximum(s[:, 1], s[:, 2]))
    x0 = np.maximum(toint(np.floor(smin(sx))) - 1, 0); x1 = np.minimum(toint(np.ceil(smax(sx))) + 1, R)
    y0 = np.maximum(toint(np.floor(smin(sy))) - 1, 0); y1 = np.minimum(toint(np.ceil(smax(sy))) + 1, R)
    alive &= (x1 > x0) & (y1 > y0)
    EA, EB, EC = [], [], []
    for i in range(3):
      ja, jb = (i + 1) % 3, (i + 2) % 3
      EA.append((sy[:, ja] - sy[:, jb]) * sgn)
      EB.append((sx[:, jb] - sx[:, ja]) * sgn)
      EC.append((sx[:, ja] * sy[:, jb] - sx[:, jb] * sy[:, ja]) * sgn)
    EA, EB, EC = np.stack(EA, 1), np.stack(EB, 1), np.stack(EC, 1)            # [T,3]
    tl = (EA > 0) | ((EA == 0) & (EB > 0))
    cons_margin = half * (np.abs(EA) + np.abs(EB))
    inv_tot = one / np.abs(area)

    ids = np.nonzero(alive)[0]
    bw = (x1 - x0)[ids]
    npix = bw * (y1 - y0)[ids]
    flat = out.reshape(M, -1)
    start = 0
    while start < len(ids):
      # greedy chunk of triangles with <= chunk_pixels bounding-box pixels (at least one triangle)
      cs = np.cumsum(npix[start:])
      n = max(1, int(np.searchsorted(cs, chunk_pixels, side="right")))
      sel, sbw, snp = ids[start:start + n], bw[start:start + n], npix[start:start + n]
      start += n
      tix = np.repeat(np.arange(len(sel)), snp)                       # pixel -> local triangle
      q = np.arange(int(snp.sum())) - np.repeat(np.cumsum(snp) - snp, snp)
      tg = sel[tix]
      PX = (x0[tg] + q % sbw[tix]).astype(f) + half
      PY = (y0[tg] + q // sbw[tix]).astype(f) + half
      inside = np.ones(len(tg), bool)
      l = []
      for i in range(3):
        E = (EA[tg, i] * PX + EB[tg, i] * PY) + EC[tg, i]
        l.append(E * inv_tot[tg])
        if conservative_rasterization:
          inside &= (E + cons_margin[tg, i]) >= 0
        else:
          inside &= (E > 0) | ((E == 0) & tl[tg, i])
      zn = (l[0] * sz[tg, 0] + l[1] * sz[tg, 1]) + l[2] * sz[tg, 2]
      inside &= (zn >= -one) & (zn <= one)
      tg = tg[inside]
      l = [li[inside] for li in l]
      pos = [(l[0] * v[tg, 0, r] + l[1] * v[tg, 1, r]) + l[2] * v[tg, 2, r] for r in range(3)]
      px, py, pz = pos
      # voxelize.frag:36-40
      ok = (px >= 0) & (py >= 0) & (pz >= 0) & (px < Wf) & (py < Hf) & (pz < Df)
      tg, px, py, pz = tg[ok], px[ok], py[ok], pz[ok]
      if vs <= 0:                                            # voxelize.frag:42-47
        cx, cy, cz = (np.floor(c).astype(np.int64) for c in (px, py, pz))
      else:                                                  # voxelize.frag:48-56
        def sub(c):
          vv = np.floor(c * f(vs)).astype(np.int64) + vs // 2
          return 2 * (vv // vs) + ((vv % vs) == vs - 1).astype(np.int64)
        cx, cy, cz = sub(px), sub(py), sub(pz)
      ok = (cx < Wg) & (cy < Hg) & (cz < Dg)
      flat[shape_index[tg[ok]], ((cz * Hg + cy) * Wg + cx)[ok]] = 1
  return out


def get_sub_grid_centers(grid: np.ndarray) -> np.ndarray:
  """voxelization.py:167-182."""
  g = grid[:, 1:, 1:, 1:]
  b, d, h, w = g.shape
  g = g.reshape(b, d // 2, 2, h // 2, 2, w // 2, 2)
  return g[:, :, 0, :, 0, :, 0]


def merge_labels(meshes_grid: np.ndarray, num_meshes: Sequence[int],
                 labels: Sequence[Sequence[int]]) -> np.ndarray:
  """batched_example.py:186-196: per scene max_m(label_m * grid_m) -> int32 (Q11)."""
  out = []
  off = 0
  for b, nm in enumerate(num_meshes):
    lab = np.asarray(labels[b], np.float32)[:, None, None, None]
    out.append((lab * meshes_grid[off:off + nm]).max(0).astype(np.int32))
    off += nm
  return np.stack(out, 0)


def transform_mesh(mesh: Tensor, matrix: Tensor) -> Tensor:
  """transformations.py:139-169 with vertices_are_points=True: pad w=1, `einsum("bnm,bvm->bvn")`
  (transform_points_homogeneous :108-136), divide by w."""
  shape = mesh.shape
  pts = t.constant_pad_nd(mesh.reshape(1, -1, 3).to(t.float32), [0, 1], value=1.0)
  res = t.einsum("bnm,bvm->bvn", matrix.reshape(1, 4, 4).to(t.float32), pts)
  return (res[..., :3] / res[..., 3:4]).reshape(shape)


def batch_vertices(examples) -> Tensor:
  """The geometry of batched_example.batch (batched_example.py:68-95): examples = [(mesh_vertices [T,3,3],
  mesh_num_tri [M], view_transform [4,4], o2w_transforms [M,4,4])]; every mesh goes to view space through
  o2v = w2v . o2w; all meshes of all scenes are concatenated."""
  out = []
  for verts, num_tri, w2v, o2ws in examples:
    off = 0
    for nt, o2w in zip(num_tri, o2ws):
      nt = int(nt)
      out.append(transform_mesh(verts[off:off + nt], t.matmul(w2v, o2w)))
      off += nt
  return t.cat(out, 0)


def view2voxel_matrices(offset: Tensor, resolution) -> Tensor:
  """batched_example.py:153-163: translate(off-0.5) @ scale(m,m,m), m=max(D,H,W) (Q9)."""
  m = float(max(resolution))
  return t.matmul(translate(offset - 0.5), scale([m, m, m]))


# ----------------------------------------------------------------------------
# Deterministic weight initialiser shared by tests / bench / golden generator
# ----------------------------------------------------------------------------
def param_specs(num_classes: int = 2, latent: int = 64, skip_fraction: float = 0.75):
  """Ordered (key, shape, kind) list with the reference's state_dict keys
  (probe of core_net.CoreNet.state_dict(); import_resnet50_checkpoint.py:27-400)."""
  specs: List[Tuple[str, Tuple[int, ...], str]] = []
  def conv(p, co, ci, *k):
    specs.append((p + "weight", (co, ci) + tuple(k), "conv"))
    specs.append((p + "bias", (co,), "bias"))
  def bn(p, c):
    specs.append((p + "weight", (c,), "bn_w"))
    specs.append((p + "bias", (c,), "bn_b"))
    specs.append((p + "running_mean", (c,), "rm"))
    specs.append((p + "running_var", (c,), "rv"))
    specs.append((p + "num_batches_tracked", (), "nbt"))
  e = "encoder."
  conv(e + "stage1.conv.", 64, 3, 7, 7)
  bn(e + "stage1_part2.bn.", 64)
  cin = 64
  for name, blocks, f in (("stage2", "abc", (64, 64, 256)), ("stage3", "abcd", (128, 128, 512)),
                          ("stage4", "abcdef", (256, 256, 1024)), ("stage5", "abc", (512, 512, 2048))):
    for bl in blocks:
      p = f"{e}{name}.{bl}."
      conv(p + "op_a.conv.", f[0], cin, 1, 1); bn(p + "op_a.bn.", f[0])
      conv(p + "op_b.conv.", f[1], f[0], 3, 3); bn(p + "op_b.bn.", f[1])
      conv(p + "op_c.conv.", f[2], f[1], 1, 1); bn(p + "op_c.bn.", f[2])
      if bl == "a":
        conv(p + "shortcut.conv.", f[2], cin, 1, 1); bn(p + "shortcut.bn.", f[2])
      cin = f[2]
  d = "decoder."
  specs.append((d + "stage_0.weight", (latent, 2048), "conv"))
  specs.append((d + "stage_0.bias", (latent,), "bias"))
  bn(d + "stage_1.b1.", latent + 3)
  specs.append((d + "stage_1.t1.weight", (latent + 3, 256, 4, 4, 4), "convT"))
  specs.append((d + "stage_1.t1.bias", (256,), "bias"))
  chans = {2: (256, 256, 128, 3, 3), 3: (None, 128, 64, 5, 7), 4: (None, 64, 32, 5, 7),
           5: (None, 32, 16, 5, 7), 6: (None, 16, num_classes, 5, 7)}
  src_c = {2: 2048, 3: 1024, 4: 512, 5: 256}
  cin = 256
  for st in range(2, 7):
    _, cmid, cout, k1, k2 = chans[st]
    p = f"{d}stage_{st}."
    bn(p + "b1.", cin)
    conv(p + "c1.", cmid, cin, k1, k1, k1)
    bn(p + "b2.", cmid)
    specs.append((p + "t1.weight", (cmid, cout, k2, k2, k2), "convT"))
    specs.append((p + "t1.bias", (cout,), "bias"))
    if st in src_c:
      sk = round(cout * skip_fraction)
      conv(f"{d}rt_skip_{st}.compress_channels.", sk, src_c[st] + 3, 1, 1)
      cin = cout + sk
  return specs


def make_state(seed: int = 0, num_classes: int = 2, nbt: int = 0,
               perturb_bn: bool = True, logit_scale: float = 1.0) -> State:
  """Deterministic He-normal conv weights; BN gamma/beta/running stats mildly
  perturbed (so that parity tests exercise them); reproducible on any host
  from the seed alone (torch CPU generator)."""
  g = t.Generator().manual_seed(seed)
  sd: State = {}
  for key, shape, kind in param_specs(num_classes):
    if kind in ("conv", "convT"):
      if kind == "conv":
        fan_in = int(np.prod(shape[1:]))
      elif "stage_1" in key:       # 1^3 -> 4^3: one tap per output voxel
        fan_in = shape[0]
      else:                        # stride 2: ~k^3/8 taps reach each output voxel
        fan_in = shape[0] * max(1, int(np.prod(shape[2:])) // 8)
      std = math.sqrt(2.0 / max(fan_in, 1))
      sd[key] = t.randn(shape, generator=g) * std
    elif kind == "bias":
      sd[key] = t.randn(shape, generator=g) * 0.05
    elif kind == "bn_w":
      sd[key] = 1.0 + (t.randn(shape, generator=g) * 0.1 if perturb_bn else t.zeros(shape))
    elif kind in ("bn_b", "rm"):
      sd[key] = t.randn(shape, generator=g) * 0.1 if perturb_bn else t.zeros(shape)
    elif kind == "rv":
      sd[key] = 1.0 + (t.rand(shape, generator=g) * 0.5 if perturb_bn else t.zeros(shape))
    elif kind == "nbt":
      sd[key] = t.tensor(nbt, dtype=t.int64)
  if logit_scale != 1.0:
    # random-init eval-mode logits reach |x| ~ 1.5e4 and saturate the softmax; a scaled last layer keeps them
    # O(1), so that probabilities (super-resolution fixtures) are a well-conditioned function of the logits
    sd["decoder.stage_6.t1.weight"] = sd["decoder.stage_6.t1.weight"] * logit_scale
    sd["decoder.stage_6.t1.bias"] = sd["decoder.stage_6.t1.bias"] * logit_scale
  return sd


def synthetic_batch(batch: int, seed: int = 0, num_classes: int = 2,
                    resolution=(128, 128, 128), image_hw=(256, 256)):
  """SURVEY 8(d) synthetic inputs: seeded uint8 image, canonical camera,
  v2s = camera @ scale(1/128), offset 0.5, GT = analytic balls."""
  g = t.Generator().manual_seed(1000 + seed)
  image = t.randint(0, 256, (batch, 3) + tuple(image_hw), generator=g, dtype=t.uint8)
  m = float(max(resolution))
  v2s = (canonical_camera() @ scale([1 / m] * 3))[None].expand(batch, 4, 4).contiguous()
  offset = t.full((batch, 3), 0.5)
  D, H, W = resolution
  zz, yy, xx = t.meshgrid(t.arange(D), t.arange(H), t.arange(W), indexing="ij")
  grid = t.zeros((batch, D, H, W), dtype=t.int64)
  nballs = 1 if num_classes == 2 else 3
  for b in range(batch):
    for k in range(nballs):
      cx = (0.5 + 0.22 * (k - (nballs - 1) / 2)) * W
      r = (0.3 if nballs == 1 else 0.1) * W
      ball = ((xx + 0.5 - cx) ** 2 + (yy + 0.5 - 0.5 * H) ** 2 + (zz + 0.5 - 0.5 * D) ** 2) <= r * r
      grid[b][ball] = 1 if num_classes == 2 else (1 + (3 * b + k) % (num_classes - 1))
  return image, v2s, offset, grid


# ----------------------------------------------------------------------------
# super-resolution inference (super_resolution.py:46-129)
def super_resolution_offsets(m: int, grid_offsets: t.Tensor) -> t.Tensor:
  """super_resolution.py:66-90: f32[m^3, B, 3]; index n=(iz*m+iy)*m+ix -> ((ix,iy,iz)+grid_offset)/m."""
  zz, yy, xx = t.meshgrid([t.arange(m)] * 3, indexing="ij")
  offsets = (t.stack([xx, yy, zz], -1) / m).reshape([-1, 3])
  return offsets[:, None] + grid_offsets[None, :] / m


def super_resolution(sd, image_u8, camera_transform, view_to_voxel, grid_offsets, m: int):
  """super_resolution.py:92-126 with the model evaluated in eval mode: pmf f32[B, C, mD, mH, mW].
  One full forward per offset, exactly like the reference."""
  native = super_resolution_offsets(m, grid_offsets)
  v2v = view_to_voxel @ scale([1.0 / m] * 3)
  v2s = camera_transform @ v2v.inverse()
  pm = []
  for off in native:
    pm.append(corenet_forward(sd, image_u8, v2s, off, training=False).softmax(dim=1))
  pm = t.stack(pm, 0)
  _, B, C, d, h, w = pm.shape
  pm = pm.reshape([m, m, m, B, C, d, h, w]).permute([3, 4, 5, 0, 6, 1, 7, 2])
  return pm.reshape([B, C, m * d, m * h, m * w])
