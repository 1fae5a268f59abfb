"""Maps the reference's Conv2d / Conv3d / ConvTranspose3d layers onto the
stride-1 window correlation implemented by crn_conv_fwd / crn_conv_wgrad.

Every layer gets two geometries:
  fwd   : y = corr(T(x), Wf)         (wgrad uses the same geometry)
  dgrad : dx = corr(dy, Wd)
each with a window (kd,kh,kw), a low-side padding and an int32 *pack index*
that gathers the packed weight [Cin_logical][taps][Npad] out of the flat
parameter buffer (index -1 = structural zero).  The forward pack index doubles
as the scatter map that un-packs weight gradients (it is injective).

Reference layers: resnet50.py:62-69,95-107,122-124 (Conv2d incl. the stride-2
1x1 and the 7x7/2 stem), reconstruction_decoder.py:52-95 (Conv3d,
ConvTranspose3d k=3/7 stride 2, and the 1^3->4^3 stage_1), and
ray_traced_skip_connection.py:38 (1x1 compress).

Transposed convolution, stride 2 (per dimension, o = 2 i - p + k):
  write o = 2 q + r.  Forward: i = q + d with k = -2 d + r + p, so the layer is a
  stride-1 correlation over d in [dmin, dmax] producing 8*Cout "parity" channels
  (rd, rh, rw, n) -- parity major -- that the kernel's output view scatters as a pixel
  shuffle.  For k = 7 one parity uses 4 of the window's taps per dimension and the other
  3 (343 = (4+3)^3 real taps in a 8 x 4^3 = 512 window): the per-parity tap boxes let the
  kernels skip the structural zeros.
  Data gradient: q = i + e with k = 2 e + r + p: a stride-1 correlation of the
  space-to-depth view of dy.  No zero-insertion, no col2im, every MFMA row dense.
"""
from __future__ import annotations

import dataclasses
from typing import Tuple

import numpy as np


def pad16(n: int) -> int:
  return (n + 15) // 16 * 16


@dataclasses.dataclass
class Geom:
  window: Tuple[int, int, int]
  pad_lo: Tuple[int, int, int]
  index: np.ndarray      # int32 [Cin_logical * taps * Npad], -1 = 0
  cin: int               # logical input channels
  nout: int              # logical output channels
  npad: int
  # structural zeros of transposed convolutions: the logical output (n_boxes) or input (c_boxes)
  # channels form len(boxes) equal contiguous groups (the 8 parities); group g only has non-zero
  # weights for taps inside the box (d0, d1, h0, h1, w0, w1) (half-open), see crnTapBoxes
  n_boxes: Tuple = ()
  c_boxes: Tuple = ()

  @property
  def taps(self) -> int:
    return self.window[0] * self.window[1] * self.window[2]


def _k3(shape_tail):
  k = tuple(shape_tail)
  return (1,) * (3 - len(k)) + k


def conv_fwd(wshape, padding) -> Geom:
  """nn.Conv{2,3}d(stride 1): W[n, c, t]."""
  N, Cc = wshape[:2]
  k = _k3(wshape[2:])
  T = k[0] * k[1] * k[2]
  npad = pad16(N)
  idx = np.full((Cc, T, npad), -1, np.int64)
  n = np.arange(N)[None, None, :]
  c = np.arange(Cc)[:, None, None]
  tt = np.arange(T)[None, :, None]
  idx[:, :, :N] = (n * Cc + c) * T + tt
  p = _k3((padding,) * (len(wshape) - 2)) if np.isscalar(padding) else _k3(padding)
  if len(wshape) == 4:
    p = (0,) + tuple(p[1:])
  return Geom(k, tuple(p), idx.reshape(-1).astype(np.int32), Cc, N, npad)


def conv_dgrad(wshape, padding) -> Geom:
  """dx = corr(dy, flip(W)^T) with pad k-1-p."""
  N, Cc = wshape[:2]
  k = _k3(wshape[2:])
  T = k[0] * k[1] * k[2]
  npad = pad16(Cc)
  idx = np.full((N, T, npad), -1, np.int64)
  n = np.arange(N)[:, None, None]
  c = np.arange(Cc)[None, None, :]
  tt = np.arange(T)[None, :, None]
  idx[:, :, :Cc] = (n * Cc + c) * T + (T - 1 - tt)
  p = _k3((padding,) * (len(wshape) - 2)) if np.isscalar(padding) else _k3(padding)
  if len(wshape) == 4:
    p = (0,) + tuple(p[1:])
  pad = tuple(kk - 1 - pp for kk, pp in zip(k, p))
  return Geom(k, pad, idx.reshape(-1).astype(np.int32), N, Cc, npad)


def _convt_ranges(ks: int, p: int):
  dmin = -((ks - 1 - p) // 2)          # ceil((p - ks + 1) / 2)
  dmax = (1 + p) // 2
  emin = (0 - p) // 2                  # floor
  emax = (ks - 1 - p) // 2
  return dmin, dmax, emin, emax


def _parity_boxes(valid_r_w: np.ndarray):
  """valid_r_w[r, w]: is window position w a real tap for parity r (one dimension)?  -> the 8 tap boxes
  (d0,d1,h0,h1,w0,w1) in (rd, rh, rw) order.  The real taps of one parity are contiguous."""
  rng = []
  for r in range(2):
    pos = np.nonzero(valid_r_w[r])[0]
    assert len(pos) > 0 and (np.diff(pos) == 1).all()
    rng.append((int(pos[0]), int(pos[-1]) + 1))
  return tuple(rng[rd] + rng[rh] + rng[rw] for rd in range(2) for rh in range(2) for rw in range(2))


def convt_fwd(wshape, padding: int) -> Geom:
  """nn.ConvTranspose3d(stride 2, output_padding 1): Wt[c, n, kd, kh, kw]."""
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  dmin, dmax, _, _ = _convt_ranges(ks, padding)
  nw = dmax - dmin + 1
  npad = pad16(N * 8)
  wq = np.arange(nw)
  r = np.arange(2)
  kk = -2 * (wq[:, None] + dmin) + r[None, :] + padding       # [nw, 2] kernel index per dim
  valid1 = (kk >= 0) & (kk < ks)
  c = np.arange(Cc).reshape(Cc, 1, 1, 1, 1, 1, 1, 1)
  n = np.arange(N).reshape(1, 1, 1, 1, N, 1, 1, 1)
  kd = kk.reshape(1, nw, 1, 1, 1, 2, 1, 1); vd = valid1.reshape(1, nw, 1, 1, 1, 2, 1, 1)
  kh = kk.reshape(1, 1, nw, 1, 1, 1, 2, 1); vh = valid1.reshape(1, 1, nw, 1, 1, 1, 2, 1)
  kw = kk.reshape(1, 1, 1, nw, 1, 1, 1, 2); vw = valid1.reshape(1, 1, 1, nw, 1, 1, 1, 2)
  flat = (((c * N + n) * ks + kd) * ks + kh) * ks + kw
  flat = np.where(vd & vh & vw, flat, -1)                      # [C, nw,nw,nw, N, 2,2,2]
  flat = flat.transpose(0, 1, 2, 3, 5, 6, 7, 4)                # parity major: [C, nw,nw,nw, 2,2,2, N]
  idx = np.full((Cc, nw ** 3, npad), -1, np.int64)
  idx[:, :, :N * 8] = flat.reshape(Cc, nw ** 3, N * 8)
  return Geom((nw,) * 3, (-dmin,) * 3, idx.reshape(-1).astype(np.int32), Cc, N * 8, npad,
              n_boxes=_parity_boxes(valid1.T))


def convt_dgrad(wshape, padding: int) -> Geom:
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  _, _, emin, emax = _convt_ranges(ks, padding)
  nw = emax - emin + 1
  npad = pad16(Cc)
  wq = np.arange(nw)
  r = np.arange(2)
  kk = 2 * (wq[None, :] + emin) + r[:, None] + padding         # [2, nw]
  valid1 = (kk >= 0) & (kk < ks)
  n = np.arange(N).reshape(N, 1, 1, 1, 1, 1, 1, 1)
  c = np.arange(Cc).reshape(1, 1, 1, 1, 1, 1, 1, Cc)
  kd = kk.reshape(1, 2, 1, 1, nw, 1, 1, 1); vd = valid1.reshape(1, 2, 1, 1, nw, 1, 1, 1)
  kh = kk.reshape(1, 1, 2, 1, 1, nw, 1, 1); vh = valid1.reshape(1, 1, 2, 1, 1, nw, 1, 1)
  kw = kk.reshape(1, 1, 1, 2, 1, 1, nw, 1); vw = valid1.reshape(1, 1, 1, 2, 1, 1, nw, 1)
  flat = (((c * N + n) * ks + kd) * ks + kh) * ks + kw
  flat = np.where(vd & vh & vw, flat, -1)                      # [N,2,2,2, nw,nw,nw, C]
  flat = flat.transpose(1, 2, 3, 0, 4, 5, 6, 7)                # parity major: [2,2,2,N, nw,nw,nw, C]
  idx = np.full((N * 8, nw ** 3, npad), -1, np.int64)
  idx[:, :, :Cc] = flat.reshape(N * 8, nw ** 3, Cc)
  return Geom((nw,) * 3, (-emin,) * 3, idx.reshape(-1).astype(np.int32), N * 8, Cc, npad,
              c_boxes=_parity_boxes(valid1))


def stem_fwd(wshape=(64, 3, 7, 7), padding: int = 3) -> Geom:
  """ZeroPad2d(3) + Conv2d(7, stride 2) on the 2x2 space-to-depth view of the
  image: i = 2 o - p + k = 2 q + r  ->  q = o + d, k = 2 d + r + p."""
  N, Cc, ks, _ = wshape
  emin, emax = (0 - padding) // 2, (ks - 1 - padding) // 2
  nw = emax - emin + 1
  npad = pad16(N)
  wq = np.arange(nw); r = np.arange(2)
  kk = 2 * (wq[None, :] + emin) + r[:, None] + padding         # [2, nw]
  valid1 = (kk >= 0) & (kk < ks)
  c = np.arange(Cc).reshape(Cc, 1, 1, 1, 1, 1)
  n = np.arange(N).reshape(1, 1, 1, 1, 1, N)
  kh = kk.reshape(1, 2, 1, nw, 1, 1); vh = valid1.reshape(1, 2, 1, nw, 1, 1)
  kw = kk.reshape(1, 1, 2, 1, nw, 1); vw = valid1.reshape(1, 1, 2, 1, nw, 1)
  flat = ((n * Cc + c) * ks + kh) * ks + kw
  flat = np.where(vh & vw, flat, -1)                           # [C,2,2, nw,nw, N]
  idx = np.full((Cc * 4, nw * nw, npad), -1, np.int64)
  idx[:, :, :N] = flat.reshape(Cc * 4, nw * nw, N)
  return Geom((1, nw, nw), (0, -emin, -emin), idx.reshape(-1).astype(np.int32), Cc * 4, N, npad)


def convt_1to4_fwd(wshape) -> Geom:
  """stage_1: ConvTranspose3d(k=4) from a 1^3 grid == a dense layer onto the
  (n, kd, kh, kw) logical channels of the 4^3 output (reconstruction_decoder.py:52-54)."""
  Cc, N = wshape[0], wshape[1]
  kv = int(np.prod(wshape[2:]))
  nn = N * kv
  npad = pad16(nn)
  idx = np.full((Cc, 1, npad), -1, np.int64)
  idx[:, 0, :nn] = np.arange(Cc)[:, None] * nn + np.arange(nn)[None, :]
  return Geom((1, 1, 1), (0, 0, 0), idx.reshape(-1).astype(np.int32), Cc, nn, npad)


def convt_1to4_dgrad(wshape) -> Geom:
  Cc, N = wshape[0], wshape[1]
  kv = int(np.prod(wshape[2:]))
  nn = N * kv
  npad = pad16(Cc)
  idx = np.full((nn, 1, npad), -1, np.int64)
  idx[:, 0, :Cc] = np.arange(Cc)[None, :] * nn + np.arange(nn)[:, None]
  return Geom((1, 1, 1), (0, 0, 0), idx.reshape(-1).astype(np.int32), nn, Cc, npad)


def bias_index(n_ref: int, repeat: int, npad: int, parity_major: bool = False) -> np.ndarray:
  """Packed bias: logical channel j takes bias[j // repeat] ((n, sub-position) order) or, with
  parity_major ((sub-position, n) order: transposed convolutions), bias[j % n_ref]."""
  idx = np.full((npad,), -1, np.int64)
  j = np.arange(n_ref * repeat)
  idx[:n_ref * repeat] = (j % n_ref) if parity_major else (j // repeat)
  return idx.astype(np.int32)


def tile_index(parts):
  """8x8 tiling of packed index arrays for crn_copy_tiles_f32.
  parts: list of (packed_offset, index int array [rows*cols] with -1 = structural zero, cols, group_rows):
  tiles never straddle a group of `group_rows` consecutive rows (the taps of one output channel in a
  data-gradient pack), inside which the index map of a plain conv is affine.
  Returns (desc int32 [n,6], mask uint64 [n], explicit int32 [m]); tiles without any element are dropped."""
  descs, masks, exs = [], [], []
  ex_off = 0
  ar = np.arange(8)
  for off, idx, cols, grows in parts:
    idx = np.asarray(idx, np.int64)
    rows = idx.size // cols
    grows = rows if not grows else grows
    assert rows % grows == 0
    ng = rows // grows
    R8, C8 = (grows + 7) // 8, (cols + 7) // 8
    pad = np.full((ng, R8 * 8, C8 * 8), -1, np.int64)
    pad[:, :grows, :cols] = idx.reshape(ng, grows, cols)
    tl = pad.reshape(ng, R8, 8, C8, 8).transpose(0, 1, 3, 2, 4).reshape(ng * R8 * C8, 8, 8)
    valid = tl >= 0
    keep = valid.any((1, 2))
    tl, valid = tl[keep], valid[keep]
    tix = np.nonzero(keep)[0]
    tg, rem = np.divmod(tix, R8 * C8)
    tr, tc = np.divmod(rem, C8)
    base = tl[:, 0, 0]
    sr = tl[:, 1, 0] - base
    sc = tl[:, 0, 1] - base
    # single-row / single-column tiles: the missing stride is irrelevant
    one_row = ~valid[:, 1:, :].any((1, 2)); one_col = ~valid[:, :, 1:].any((1, 2))
    sr = np.where(one_row, 0, sr); sc = np.where(one_col, 0, sc)
    pred = base[:, None, None] + ar[None, :, None] * sr[:, None, None] + ar[None, None, :] * sc[:, None, None]
    affine = valid[:, 0, 0] & (valid[:, 1, 0] | one_row) & (valid[:, 0, 1] | one_col) & (~valid | (tl == pred)).all((1, 2))
    affine &= (np.abs(sr) < 2 ** 31) & (np.abs(sc) < 2 ** 31)
    n = tl.shape[0]
    d = np.zeros((n, 6), np.int64)
    d[:, 0] = off + (tg * grows + tr * 8) * cols + tc * 8
    d[:, 1] = cols
    d[:, 2] = np.where(affine, base, 0); d[:, 3] = np.where(affine, sr, 0); d[:, 4] = np.where(affine, sc, 0)
    nex = int((~affine).sum())
    d[:, 5] = -1
    d[~affine, 5] = ex_off + np.arange(nex) * 64
    exs.append(np.where(valid[~affine], tl[~affine], 0).reshape(-1))
    ex_off += nex * 64
    bits = (valid.reshape(n, 64).astype(np.uint64) << np.arange(64, dtype=np.uint64)[None, :]).sum(1, dtype=np.uint64)
    descs.append(d); masks.append(bits)
  desc = np.concatenate(descs) if descs else np.zeros((0, 6), np.int64)
  assert np.abs(desc).max(initial=0) < 2 ** 31
  ex = np.concatenate(exs) if exs else np.zeros((0,), np.int64)
  return desc.astype(np.int32), (np.concatenate(masks) if masks else np.zeros((0,), np.uint64)), ex.astype(np.int32)


MAT_LDS = 8448      # floats of LDS per block of crn_copy_mats_f32 (csrc/misc_ops.hip)


def mat_index(parts):
  """Blocks for crn_copy_mats_f32 out of the same `parts` as tile_index: a part whose index map is one affine
  rectangle per row group -- idx[g*grows + r][c] = base + g*gs + r*rs + c*cs on r < R, c < N, -1 elsewhere -- with
  rs = +-1 (forward layouts: the packed matrix is the transpose of the reference one; data-gradient layouts: one small
  flipped transpose per output channel) or cs = 1 (1x1 data gradients: a strided copy) is cut into G x A x B blocks,
  b along the reference layout's contiguous axis.  Returns (desc int32 [k, 16] incl. the kernel's three reciprocals, the parts that do not fit: transposed
  convolutions, the stem, repeated biases -- they stay on the 8x8 tiles)."""
  descs, rest = [], []
  for part in parts:
    off, idx, cols, grows = part
    idx2 = np.asarray(idx, np.int64).reshape(-1, cols)
    rows = idx2.shape[0]
    gr = grows or rows
    ng = rows // gr
    idx3 = idx2.reshape(ng, gr, cols)
    v = idx3 >= 0
    R, N = int(v[0, :, 0].sum()), int(v[0, 0, :].sum())
    ok = R > 0 and N > 0 and R * N > 1 and bool(v[:, :R, :N].all()) and int(v.sum()) == ng * R * N
    if ok:
      base = int(idx3[0, 0, 0])
      rs = int(idx3[0, 1, 0]) - base if R > 1 else 0
      cs = int(idx3[0, 0, 1]) - base if N > 1 else 0
      gs = int(idx3[1, 0, 0]) - base if ng > 1 else 0
      pred = (base + np.arange(ng)[:, None, None] * gs + np.arange(R)[None, :, None] * rs + np.arange(N)[None, None, :] * cs)
      ok = bool((pred == idx3[:, :R, :N]).all()) and (abs(rs) == 1 or cs == 1)
    if not ok:
      rest.append(part)
      continue
    if abs(rs) == 1 and R > 1:
      # b = row (reversed when rs == -1), a = column
      nb, na = R, N
      f0 = base if rs == 1 else base - (R - 1)
      p0 = off if rs == 1 else off + (R - 1) * cols
      fa, pa, pb = cs, 1, (cols if rs == 1 else -cols)
    else:
      # b = column, a = row: both sides contiguous along b
      nb, na = N, R
      f0, p0 = base, off
      fa, pa, pb = rs, cols, 1
    fg, pg = gs, gr * cols
    bstep = nb if nb <= 128 else 64
    for b0 in range(0, nb, bstep):
      B = min(bstep, nb - b0)
      for a0 in range(0, na, 64):
        A = min(64, na - a0)
        gstep = max(1, MAT_LDS // (A * (B | 1))) if (B == nb and ng > 1) else 1
        g0 = np.arange(0, ng, gstep)
        G_ = np.minimum(gstep, ng - g0)
        d = np.zeros((len(g0), 16), np.int64)
        d[:, 0], d[:, 1], d[:, 2] = A, B, G_
        d[:, 3] = f0 + g0 * fg + a0 * fa + b0
        d[:, 4], d[:, 5] = fa, fg
        d[:, 6] = p0 + g0 * pg + a0 * pa + b0 * pb
        d[:, 7], d[:, 8], d[:, 9] = pa, pb, pg
        magic = lambda q: 0 if q == 1 else ((1 << 32) + q - 1) // q
        d[:, 10], d[:, 11], d[:, 12] = magic(A * B), magic(B), magic(A)
        descs.append(d)
  desc = np.concatenate(descs) if descs else np.zeros((0, 16), np.int64)
  assert np.abs(desc[:, :10]).max(initial=0) < 2 ** 31 and (desc[:, 2] * desc[:, 0] * (desc[:, 1] | 1) <= MAT_LDS).all()
  return desc.astype(np.uint32).view(np.int32), rest


# ---- operand blocks of the encoder engine (csrc/conv_e2d.hip, include/corenet_hip.h crn_bf3_operands) ----
def operand_eligible(g: Geom) -> bool:
  """Layers crn_conv2d_bf3 covers: 1x1 / 3x3 windows over 2-D images, Cin % 32 == 0, Cout % 64 == 0 == Npad."""
  return (g.window in ((1, 1, 1), (1, 3, 3)) and g.cin % 32 == 0 and g.nout % 64 == 0 and g.npad == g.nout
          and not g.n_boxes and not g.c_boxes)


def operand_entries(g: Geom) -> int:
  """32-byte entries of one layer's operand blocks."""
  return (g.cin // 32) * g.taps * (g.npad // 16) * 64


def slab_entries(g: Geom) -> int:
  """32-byte entries of one layer in slab order (decoder engine, crn_conv_fwd_bf3_slabs)."""
  khw = g.window[1] * g.window[2]
  return ((g.cin + 7) // 8) * g.window[0] * ((khw + 3) // 4 * 4) * g.npad


def operand_table(layers):
  """layers: [(first float in the packed buffer, first entry in the operand buffer, Geom, slab order?)] ->
  (int64 [n, 7] rows (src, dst, Cin, T, Npad, first workgroup, KHW or 0), total workgroups of 256 entries)."""
  rows, blocks = [], 0
  for layer in layers:
    src, dst, g = layer[:3]
    slab = len(layer) > 3 and layer[3]
    rows.append((src, dst, g.cin, g.taps, g.npad, blocks, g.window[1] * g.window[2] if slab else 0))
    blocks += ((slab_entries(g) if slab else operand_entries(g)) + 255) // 256
  return np.asarray(rows, dtype=np.int64).reshape(-1, 7), blocks


# ---------------------------------------------------------------------------------------------------------------
# Weight images of the parity-walk kernels of decoder stage_6.t1 (csrc/convt_par.hip, ConvTranspose3d 16 -> C, k 7, stride 2,
# padding 3; reconstruction_decoder.py:89-95).  A table row = 8 source indices into the flat parameter slab (-1: zero) and
# the two destination entries (16-byte units) of their bf16 hi / lo parts: crn_bf3_gather_image.  Image = the steps of the
# kernel's walk in order, per step hi[rows][4 taps][32 columns] then lo[...]; + 4 KiB of slack (a 3-row step is followed
# by a 4-row one at most: the kernels never read past a step, the slack only keeps every piece inside the buffer).
def _ct_table(rows_of_step):
  tab, off16 = [], 0
  for rows, entry in rows_of_step:                  # rows: window rows of the step; entry(zi_h, kk, col) -> 8 indices
    hi0, lo0 = off16, off16 + rows * 128
    for zi in range(rows):
      for kk in range(4):
        for col in range(32):
          e = (zi * 4 + kk) * 32 + col
          tab.append(list(entry(zi, kk, col)) + [hi0 + e, lo0 + e])
    off16 += rows * 256
  return np.asarray(tab, np.int32), off16 * 16 + 4096


def convt_par_fwd_table(wshape, wofs: int = 0):
  """Forward walk: (rd, rh) x 8-channel chunk of the 16 inputs x window plane zd < 3 + rd; rows zh < 3 + rh; column =
  rw * 16 + n; tap k = 5 - 2 z + r per dimension (convt_fwd above)."""
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  assert Cc == 16 and ks == 7 and N <= 16
  steps = []
  for pp in range(4):
    rd, rh = pp >> 1, pp & 1
    for ch in range(2):
      for zd in range(3 + rd):
        def entry(zh, kk, col, rd=rd, rh=rh, ch=ch, zd=zd):
          rw, n = col >> 4, col & 15
          kd, kh, kw = 5 - 2 * zd + rd, 5 - 2 * zh + rh, 5 - 2 * kk + rw
          if n >= N or kw < 0:
            return [-1] * 8
          return [wofs + (((ch * 8 + j) * N + n) * ks + kd) * ks * ks + kh * ks + kw for j in range(8)]
        steps.append((3 + rh, entry))
  return _ct_table(steps)


def convt_par_dgrad_table(wshape, wofs: int = 0):
  """Data-gradient walk: (rd, rh) x 8-channel half of n x window plane z in [1 - rd, 4); rows z in [1 - rh, 4); column =
  rw * 16 + c; tap k = 2 z - 1 + r (convt_dgrad above); the 8 bf16 of an entry are 8 output channels n."""
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  assert Cc == 16 and ks == 7 and N <= 16
  steps = []
  for pp in range(4):
    rd, rh = pp >> 1, pp & 1
    for nh in range((N + 7) // 8):
      for zd in range(1 - rd, 4):
        def entry(zi, kk, col, rd=rd, rh=rh, nh=nh, zd=zd):
          rw, c = col >> 4, col & 15
          zh = 1 - rh + zi
          kd, kh, kw = 2 * zd - 1 + rd, 2 * zh - 1 + rh, 2 * kk - 1 + rw
          if kw < 0:
            return [-1] * 8
          return [(wofs + ((c * N + nh * 8 + j) * ks + kd) * ks * ks + kh * ks + kw) if nh * 8 + j < N else -1 for j in range(8)]
        steps.append((3 + rh, entry))
  return _ct_table(steps)


def convt_res_fwd_table(wshape, wofs: int = 0):
  """Two classes: the resident-weights forward kernel (convt_res_kernel<false>).  Image = its LDS layout: hi[chunk][zd][zh][tap zw][16
  columns] then lo[...]; column = ((rd * 2 + rh) * 2 + rw) * 2 + n; entry = 8 input channels of the chunk; tap k = 5 - 2 z + r."""
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  assert Cc == 16 and ks == 7 and N == 2
  tab = []
  for c in range(2):
    for zd in range(4):
      for zh in range(4):
        for kk in range(4):
          for col in range(16):
            rd, rh, rw, n = col >> 3, (col >> 2) & 1, (col >> 1) & 1, col & 1
            kd, kh, kw = 5 - 2 * zd + rd, 5 - 2 * zh + rh, 5 - 2 * kk + rw
            e = ((c * 4 + zd) * 4 + zh) * 64 + kk * 16 + col
            if min(kd, kh, kw) < 0:
              src = [-1] * 8
            else:
              src = [wofs + (((c * 8 + j) * N + n) * ks + kd) * ks * ks + kh * ks + kw for j in range(8)]
            tab.append(src + [e, 2048 + e])
  return np.asarray(tab, np.int32), 2 * 2048 * 16


def convt_res_dgrad_table(wshape, wofs: int = 0):
  """... and its data gradient (convt_res_kernel<true>): chunk = rd, the 8 channels of an entry are (rh, rw, n) of dy, column = input
  channel c; tap k = 2 z - 1 + r."""
  Cc, N, ks = wshape[0], wshape[1], wshape[2]
  assert Cc == 16 and ks == 7 and N <= 2
  tab = []
  for rd in range(2):
    for zd in range(4):
      for zh in range(4):
        for kk in range(4):
          for c in range(16):
            e = ((rd * 4 + zd) * 4 + zh) * 64 + kk * 16 + c
            src = []
            for j in range(8):
              rh, rw, n = j >> 2, (j >> 1) & 1, j & 1
              kd, kh, kw = 2 * zd - 1 + rd, 2 * zh - 1 + rh, 2 * kk - 1 + rw
              src.append(-1 if min(kd, kh, kw) < 0 or n >= N else wofs + ((c * N + n) * ks + kd) * ks * ks + kh * ks + kw)
            tab.append(src + [e, 2048 + e])
  return np.asarray(tab, np.int32), 2 * 2048 * 16
