"""Executable SPECIFICATION of the libcorenet_hip.so kernel contracts, written
with torch-CPU ops.  TEST INFRASTRUCTURE ONLY.

It mirrors corenet_amd.backend.HipBackend method for method so that the host
wiring of corenet_amd/model/engine.py (views, packed-weight tables, buffer
plumbing, the forward/backward sequence) can be validated against the oracle
in this GPU-less container.  The product never imports this module: on a GPU
box every one of these methods is a HIP kernel, and the `-m gpu` tests compare
the HIP kernels against these contracts / the oracle.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch as t
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import corenet_oracle as O   # noqa: E402
from corenet_amd.views import View        # noqa: E402


def _chan(view: View, c: int) -> t.Tensor:
  off = int(view.chan_off[c]) if view.chan_off is not None else c * view.sC
  return t.as_strided(view.storage, (view.B, view.D, view.H, view.W),
                      (view.sB, view.sD, view.sH, view.sW), view.offset + off)


def logical(view: View) -> t.Tensor:
  if view.chan_off is None:
    return t.as_strided(view.storage, (view.B, view.C, view.D, view.H, view.W),
                        (view.sB, view.sC, view.sD, view.sH, view.sW), view.offset)
  return t.stack([_chan(view, c) for c in range(view.C)], 1)


def write_logical(view: View, val: t.Tensor, accumulate=False):
  if view.chan_off is None:
    dst = t.as_strided(view.storage, (view.B, view.C, view.D, view.H, view.W),
                       (view.sB, view.sC, view.sD, view.sH, view.sW), view.offset)
    if accumulate: dst += val
    else: dst.copy_(val)
    return
  for c in range(view.C):
    dst = _chan(view, c)
    if accumulate: dst += val[:, c]
    else: dst.copy_(val[:, c])


def _transform(x, tr):
  if tr is None or tr.scale is None:
    return x
  if tr.pre_relu: x = x.relu()
  shape = [1, -1] + [1] * (x.dim() - 2)
  x = x * tr.scale.view(shape) + tr.shift.view(shape)
  if tr.post_relu: x = x.relu()
  return x


def _padded(x, window, pad_lo, out_dims):
  pads = []
  for dim in (2, 1, 0):     # F.pad order: W, H, D
    lo = pad_lo[dim]
    hi = out_dims[dim] + window[dim] - 1 - lo - x.shape[2 + dim]
    pads += [lo, hi]
  return F.pad(x, pads)     # negative pads crop


class EmuBackend:
  name = "emu"

  # -- convolution engine -----------------------------------------------------
  def conv_fwd(self, x, tr, w, npad, bias, bias_sB, y, window, pad_lo, splits=1, accumulate=False, boxes=None,
               math="fp32", wslab=None):
    # boxes only tell where the packed weights are structurally zero: no effect on the result
    if wslab is not None:        # slab order (crn_bf3_operands, KHW > 0): [chunk][zd][tap slot][n][hi 8 | lo 8]
      kd, khw = window[0], window[1] * window[2]
      tp, nch = (khw + 3) // 4 * 4, (x.C + 7) // 8
      ent = wslab.view(t.int16)[:nch * kd * tp * npad * 16].view(t.bfloat16).view(nch, kd, tp, npad, 2, 8).float()
      w = (ent[..., 0, :] + ent[..., 1, :])[:, :, :khw].permute(0, 4, 1, 2, 3).reshape(nch * 8, kd * khw, npad)[:x.C]
      w = w.reshape(-1).to(logical(x).dtype)
    xl = _transform(logical(x), tr)
    xp = _padded(xl, window, pad_lo, (y.D, y.H, y.W))
    T = window[0] * window[1] * window[2]
    wk = w.view(x.C, window[0], window[1], window[2], npad).permute(4, 0, 1, 2, 3)
    out = F.conv3d(xp, wk)[:, :y.C]
    if bias is not None:
      if bias_sB:
        out = out + bias.view(x.B, -1)[:, :y.C, None, None, None]
      else:
        out = out + bias[:y.C].view(1, -1, 1, 1, 1)
    write_logical(y, out, accumulate)

  def splitk_defer(self, on=True):
    pass                                       # a scheduling hint: no effect on results

  # encoder engine (csrc/conv_e2d.hip): operand blocks [(cb*T + t)][ntile][kk*16 + i][hi 8 | lo 8] bf16
  def bf3_operands(self, packed, table, out):
    desc, _blocks = table
    o16 = out.view(t.int16)
    for src, dst, cin, T, npad, _first, khw in desc.cpu().tolist():
      if khw:                                                                              # slab order
        kd, tp, nch = T // khw, (khw + 3) // 4 * 4, (cin + 7) // 8
        w = t.zeros(nch * 8, kd, tp, npad, dtype=t.float32)
        w[:cin, :, :khw] = packed[src:src + cin * T * npad].float().view(cin, kd, khw, npad)
        blk = w.view(nch, 8, kd, tp, npad).permute(0, 2, 3, 4, 1).contiguous()             # chunk zd tp n j
        hi = blk.to(t.bfloat16)
        lo = (blk - hi.float()).to(t.bfloat16)
        ent = t.stack([hi, lo], dim=-2).reshape(-1)
        o16[dst * 16:dst * 16 + ent.numel()] = ent.view(t.int16)
        continue
      w = packed[src:src + cin * T * npad].view(cin // 32, 4, 8, T, npad // 16, 16)       # cb kk j t ntile i
      blk = w.permute(0, 3, 4, 1, 5, 2).contiguous()                                       # cb t ntile kk i j
      hi = blk.to(t.bfloat16)
      lo = (blk - hi.float()).to(t.bfloat16)
      ent = t.stack([hi, lo], dim=-2).reshape(-1)                                          # ... [hi 8 | lo 8]
      o16[dst * 16:dst * 16 + ent.numel()] = ent.view(t.int16)

  def bf3_gather_image(self, src, table, out, host_table=None):
    tab = table.cpu().long()
    v = t.where(tab[:, :8] >= 0, src.float()[tab[:, :8].clamp(min=0)], t.zeros(()))
    hi = v.to(t.bfloat16)
    lo = (v - hi.float()).to(t.bfloat16)
    o16 = out.view(t.int16).view(-1, 8)
    o16[tab[:, 8]] = hi.view(t.int16)
    o16[tab[:, 9]] = lo.view(t.int16)

  @staticmethod
  def _ct_weights(wimg, host_table, cout):
    """The [16, cout, 7, 7, 7] weights a parity-walk image holds (hi + lo), via the table that built it (source offset 0)."""
    tab = t.as_tensor(host_table).long()
    ent = wimg.view(t.int16).view(-1, 8).view(t.bfloat16).float()
    v = ent[tab[:, 8]] + ent[tab[:, 9]]
    w = t.zeros(16 * cout * 343)
    m = tab[:, :8] >= 0
    w[tab[:, :8][m]] = v[m]
    return w.view(16, cout, 7, 7, 7)

  def convt_par_fwd(self, x, tr, wimg, bias, y, cout, host_table=None, resident=False):
    w = self._ct_weights(wimg, host_table, cout)
    xt = x.float()
    if tr is not None:
      if tr.pre_relu: xt = xt.relu()
      xt = xt * tr.scale.view(1, -1, 1, 1, 1) + tr.shift.view(1, -1, 1, 1, 1)
      if tr.post_relu: xt = xt.relu()
    b = bias[:cout].float() if bias is not None else None
    y[:, :cout] = t.nn.functional.conv_transpose3d(xt, w, b, stride=2, padding=3, output_padding=1).to(y.dtype)

  def convt_par_dgrad(self, dy, cout, wimg, dx, accumulate=False, host_table=None, resident=False):
    w = self._ct_weights(wimg, host_table, cout)
    g = t.nn.functional.conv3d(dy[:, :cout].float(), w, stride=2, padding=3)          # the adjoint of the transposed conv
    if accumulate: dx += g.to(dx.dtype)
    else: dx.copy_(g)

  def conv2d_bf3(self, x, tr, wop, npad, bias, bias_sB, y, window, pad_lo, accumulate=False):
    T = window[1] * window[2]
    n = (x.C // 32) * T * (npad // 16) * 64 * 16
    ent = wop.view(t.int16)[:n].view(t.bfloat16).view(x.C // 32, T, npad // 16, 4, 16, 2, 8).float()
    blk = ent[..., 0, :] + ent[..., 1, :]                                                  # cb t ntile kk i j
    w = blk.permute(0, 3, 5, 1, 2, 4).reshape(-1)                                          # [c][t][n]
    self.conv_fwd(x, tr, w, npad, bias, bias_sB, y, window, pad_lo, 1, accumulate)

  @t.enable_grad()        # uses autograd as a calculator; may be called from inside an autograd Function
  def conv_wgrad(self, x, tr, dy, dw, npad, window, pad_lo, zero_first=True, boxes=None, math="fp32", ximg=None):
    xl = _transform(logical(x), tr)
    xp = _padded(xl, window, pad_lo, (dy.D, dy.H, dy.W))
    dyl = logical(dy)
    wk = t.zeros(dy.C, x.C, *window, requires_grad=True, dtype=xp.dtype)
    out = F.conv3d(xp, wk)
    (g,) = t.autograd.grad(out, wk, dyl)                 # [N, C, kd,kh,kw]
    T = window[0] * window[1] * window[2]
    full = t.zeros(x.C, T, npad, dtype=xp.dtype)
    full[:, :, :dy.C] = g.permute(1, 2, 3, 4, 0).reshape(x.C, T, dy.C)
    if zero_first: dw.zero_()
    dw += full.reshape(-1)

  def copy_mats(self, src, dst, desc, reverse=False):
    """crn_copy_mats_f32 as include/corenet_hip.h words it: block (g, a, b) <-> reference fbase + g*fg + a*fa + b,
    packed pbase + g*pg + a*pa + b*pb."""
    for A, B, Gn, fbase, fa, fg, pbase, pa, pb, pg in desc.cpu().long()[:, :10].tolist():
      g = t.arange(Gn).view(-1, 1, 1); a = t.arange(A).view(1, -1, 1); b = t.arange(B).view(1, 1, -1)
      ref = (fbase + g * fg + a * fa + b).reshape(-1)
      pk = (pbase + g * pg + a * pa + b * pb).reshape(-1)
      if reverse: dst[ref] = src[pk]
      else: dst[pk] = src[ref]

  def copy_tiles(self, src, dst, tiles, reverse=False):
    if len(tiles) > 3 and tiles[3].shape[0]:
      self.copy_mats(src, dst, tiles[3], reverse)
    desc, mask, ex = [x.cpu() for x in tiles[:3]]
    n = desc.shape[0]
    if not n:
      return
    r = t.arange(8).view(1, 8, 1); c = t.arange(8).view(1, 1, 8)
    d = desc.long()
    pos = (d[:, 0].view(n, 1, 1) + r * d[:, 1].view(n, 1, 1) + c).reshape(-1)
    aff = (d[:, 2].view(n, 1, 1) + r * d[:, 3].view(n, 1, 1) + c * d[:, 4].view(n, 1, 1)).reshape(n, 64)
    lane = t.arange(64).view(1, 64)
    exi = ex.long()[(d[:, 5].clamp(min=0).view(n, 1) + lane).clamp(max=max(ex.numel() - 1, 0))] if ex.numel() else aff
    index = t.where(d[:, 5].view(n, 1) >= 0, exi, aff).reshape(-1)
    bits = ((mask.view(n, 1) >> lane) & 1).bool().reshape(-1)
    if reverse: dst[index[bits]] = src[pos[bits]]
    else: dst[pos[bits]] = src[index[bits]]

  def gather(self, src, idx, dst):
    i = idx.long()
    dst.copy_(t.where(i >= 0, src[i.clamp(min=0)], t.zeros((), dtype=src.dtype)))

  def scatter(self, src, idx, dst, accumulate=False):
    i = idx.long(); m = i >= 0
    if accumulate: dst.index_add_(0, i[m], src[m])
    else: dst[i[m]] = src[m]

  def bias_grad(self, dy, B, Cn, S, sB, db, accumulate=False):
    v = t.as_strided(dy, (B, Cn, S), (sB, S, 1), dy.storage_offset()).double().sum((0, 2)).to(dy.dtype)
    if accumulate: db += v
    else: db.copy_(v)

  # -- BatchRenorm --------------------------------------------------------------
  def bn_stats(self, x, B, Cn, S, sB, pre_relu, gamma, beta, rmean, rvar, nbt, eps, momentum,
               training, scale, shift, saved):
    if not training:
      rstd = 1.0 / (rvar + eps).sqrt()
      scale.copy_(gamma * rstd); shift.copy_(beta - gamma * rmean * rstd)
      return
    v = t.as_strided(x, (B, Cn, S), (sB, S, 1), x.storage_offset())
    if pre_relu: v = v.relu()
    vd = v.double()
    mean = vd.mean((0, 2)); var = (vd * vd).mean((0, 2)) - mean * mean
    b_mean, b_var = mean.to(x.dtype), var.clamp(min=0).to(x.dtype)
    b_std = (b_var + eps).sqrt(); run_std = (rvar + eps).sqrt()
    nt = nbt.reshape(())          # float32 schedule arithmetic, as batch_renorm.py:41-42
    d_max = float((5.0 * (nt - 5000) / (25000 - 5000)).clamp(0.0, 5.0))
    r_max = float(1.0 + (2.0 * (nt - 5000) / (40000 - 5000)).clamp(0.0, 2.0))
    r = (b_std / run_std).clamp(1 / r_max, r_max)
    d = ((b_mean - rmean) / run_std).clamp(-d_max, d_max)
    rstd = 1.0 / b_std
    scale.copy_(gamma * r * rstd); shift.copy_(beta + gamma * (d - b_mean * r * rstd))
    saved.view(4, Cn).copy_(t.stack([b_mean, rstd, r, d]))
    rvar += momentum * (b_var * Cn / (Cn - 1) - rvar)
    rmean += momentum * (b_mean - rmean)

  def bn_bwd(self, x, sB_x, dy, sB_dy, B, Cn, S, pre_relu, post_relu, gamma, scale, shift, saved,
             dx, sB_dx, dgamma, dbeta, accumulate=False, dsum=None, ndsum=0):
    xs = t.as_strided(x, (B, Cn, S), (sB_x, S, 1), x.storage_offset())
    g = t.as_strided(dy, (B, Cn, S), (sB_dy, S, 1), dy.storage_offset()).clone()
    mu, rstd, r, d = [u.view(1, Cn, 1) for u in saved.view(4, Cn)]
    xv = xs.relu() if pre_relu else xs
    if post_relu:
      g = g * ((xv * scale.view(1, Cn, 1) + shift.view(1, Cn, 1)) > 0)
    xn = (xv - mu) * rstd
    n = B * S
    s1 = g.double().sum((0, 2)); s2 = (g.double() * xn.double()).sum((0, 2))
    dg = (r.view(-1).double() * s2 + d.view(-1).double() * s1).to(x.dtype); db = s1.to(x.dtype)
    if accumulate: dgamma += dg; dbeta += db
    else: dgamma.copy_(dg); dbeta.copy_(db)
    mg = (s1 / n).to(x.dtype).view(1, Cn, 1); mgx = (s2 / n).to(x.dtype).view(1, Cn, 1)
    o = gamma.view(1, Cn, 1) * r * rstd * (g - mg - xn * mgx)
    if pre_relu: o = o * (xs > 0)
    t.as_strided(dx, (B, Cn, S), (sB_dx, S, 1), dx.storage_offset()).copy_(o)
    if dsum is not None:
      dsum[:ndsum].copy_(o[:, :ndsum].double().sum((0, 2)).to(x.dtype))

  def bn_eval_affine(self, params, buffers, table, eps, scale, shift):
    tb = table.long()
    g, bt, rm, rv = params[tb[:, 0]], params[tb[:, 1]], buffers[tb[:, 2]], buffers[tb[:, 3]]
    rstd = 1.0 / (rv + eps).sqrt()
    scale[tb[:, 4]] = g * rstd
    shift[tb[:, 4]] = bt - g * rm * rstd

  def affine_add_relu(self, x, scale, shift, r, rscale, rshift, B, Cn, S, sB_x, sB_r, y_pre, sB_pre,
                      y, sB_y, relu):
    v = t.as_strided(x, (B, Cn, S), (sB_x, S, 1), x.storage_offset())
    one = t.ones(Cn, dtype=x.dtype); zero = t.zeros(Cn, dtype=x.dtype)
    o = v * (scale if scale is not None else one).view(1, Cn, 1) + (shift if shift is not None else zero).view(1, Cn, 1)
    if r is not None:
      rv = t.as_strided(r, (B, Cn, S), (sB_r, S, 1), r.storage_offset())
      o = o + rv * (rscale if rscale is not None else one).view(1, Cn, 1) + (rshift if rshift is not None else zero).view(1, Cn, 1)
    if y_pre is not None:
      t.as_strided(y_pre, (B, Cn, S), (sB_pre, S, 1), y_pre.storage_offset()).copy_(o)
    if y is not None:
      t.as_strided(y, (B, Cn, S), (sB_y, S, 1), y.storage_offset()).copy_(o.relu() if relu else o)

  def relu_bwd_add(self, dy, y_pre, dy2, B, Cn, S, sB_dy, sB_pre, sB_dy2, dx, sB_dx):
    p = t.as_strided(y_pre, (B, Cn, S), (sB_pre, S, 1), y_pre.storage_offset())
    o = t.zeros(B, Cn, S, dtype=dx.dtype)
    if dy is not None:
      o = t.as_strided(dy, (B, Cn, S), (sB_dy, S, 1), dy.storage_offset()) * (p > 0)
    if dy2 is not None:
      o = o + t.as_strided(dy2, (B, Cn, S), (sB_dy2, S, 1), dy2.storage_offset())
    t.as_strided(dx, (B, Cn, S), (sB_dx, S, 1), dx.storage_offset()).copy_(o)

  # -- encoder odds and ends -------------------------------------------------------
  def preprocess(self, img_u8, out):
    out.copy_(O.preprocess_image_caffe(img_u8))

  def maxpool_fwd(self, x, scale, shift, B, Cn, H, W, y, argmax):
    a = (x * scale.view(1, Cn, 1, 1) + shift.view(1, Cn, 1, 1)).relu()
    ap = F.pad(a, [1, 1, 1, 1])
    v, idx = F.max_pool2d(ap, 3, 2, return_indices=True)
    ih = idx // (W + 2) - 1; iw = idx % (W + 2) - 1
    ok = (ih >= 0) & (ih < H) & (iw >= 0) & (iw < W) & (v > 0)
    y.copy_(v)
    argmax.copy_(t.where(ok, ih * W + iw, -t.ones_like(ih)).to(t.int32))

  def maxpool_bwd(self, dy, argmax, B, Cn, H, W, dx):
    dxf = t.zeros(B * Cn, H * W, dtype=dy.dtype)
    am = argmax.reshape(B * Cn, -1).long(); g = dy.reshape(B * Cn, -1)
    dxf.scatter_add_(1, am.clamp(min=0), g * (am >= 0))
    dx.copy_(dxf.view(B, Cn, H, W))

  def relu_mean_fwd(self, x_pre, B, Cn, S, sB, avg):
    v = t.as_strided(x_pre, (B, Cn, S), (sB, S, 1), x_pre.storage_offset())
    avg.copy_(v.relu().double().mean(2).to(avg.dtype))

  def relu_mean_bwd(self, x_pre, davg, B, Cn, S, sB, dx, sB_dx, accumulate=False):
    v = t.as_strided(x_pre, (B, Cn, S), (sB, S, 1), x_pre.storage_offset())
    g = (v > 0) * (davg.view(B, Cn, 1) / float(S))
    dst = t.as_strided(dx, (B, Cn, S), (sB_dx, S, 1), dx.storage_offset())
    if accumulate: dst += g
    else: dst.copy_(g)

  def linear_fwd(self, x, w, bias, B, K, N, y, ldy):
    t.as_strided(y, (B, N), (ldy, 1), y.storage_offset()).copy_(F.linear(x.view(B, K), w, bias))

  def linear_bwd(self, x, w, dy, lddy, B, K, N, dx, dw, db):
    g = t.as_strided(dy, (B, N), (lddy, 1), dy.storage_offset())
    if dx is not None: dx.view(B, K).copy_(g @ w)
    if dw is not None: dw.copy_(g.t() @ x.view(B, K))
    if db is not None: db.copy_(g.sum(0))

  def stride2_gather(self, x, y):
    y.copy_(x[:, :, ::2, ::2])

  def stride2_scatter(self, dy, dx):
    dx.zero_()
    dx[:, :, ::2, ::2] = dy

  def decoder_inputs(self, v2s, offset, scales, layer_mats, offset_out):
    B = v2s.shape[0]
    sc = t.tensor([[float(s)] * 3 + [1.0] for s in scales], dtype=layer_mats.dtype).reshape(len(scales), 1, 1, 4)
    layer_mats.copy_((v2s.to(layer_mats.dtype).reshape(1, B, 4, 4) * sc).reshape(len(scales), B, 16))
    offset_out.copy_(offset)

  def fill_offset_channels(self, x, B, sB, S, c0, offset):
    t.as_strided(x, (B, 3, S), (sB, S, 1), x.storage_offset() + c0 * S).copy_(
        offset.view(B, 3, 1).expand(B, 3, S))

  # -- ray-traced skip ------------------------------------------------------------------
  def ray_sample_fwd(self, fmap, map_sB, B, Cn, h, w, matrix, offset, out, out_sB, D, H, W,
                     map_sC=None, map_sP=1):
    sC = h * w if map_sC is None else map_sC
    m = t.as_strided(fmap, (B, Cn, h, w), (map_sB, sC, w * map_sP, map_sP), fmap.storage_offset())
    res = O.ray_sample(m, matrix.view(B, 4, 4).float(), offset.view(B, 3).float(), (D, H, W))
    t.as_strided(out, (B, Cn, D, H, W), (out_sB, D * H * W, H * W, W, 1), out.storage_offset()).copy_(res)

  def ray_sample_bwd(self, dout, dout_sB, B, Cn, D, H, W, matrix, offset, dmap, dmap_sB, h, w,
                     zero_first=True):
    g = t.as_strided(dout, (B, Cn, D, H, W), (dout_sB, D * H * W, H * W, W, 1), dout.storage_offset())
    iy, ix, keep = O.ray_sample_indices(matrix.view(B, 4, 4).float(), offset.view(B, 3).float(), (D, H, W), (w, h))
    pad = t.zeros(B, Cn, h + 2, w + 2, dtype=g.dtype)
    bb = t.arange(B)[:, None, None, None].expand_as(iy)
    gm = (g * keep[:, None]).permute(0, 2, 3, 4, 1)
    flat = pad.permute(0, 2, 3, 1).reshape(-1, Cn)
    lin = ((bb * (h + 2) + iy) * (w + 2) + ix).reshape(-1)
    flat.index_add_(0, lin, gm.reshape(-1, Cn))
    res = flat.view(B, h + 2, w + 2, Cn).permute(0, 3, 1, 2)[:, :, 1:-1, 1:-1]
    dst = t.as_strided(dmap, (B, Cn, h, w), (dmap_sB, h * w, w, 1), dmap.storage_offset())
    if zero_first: dst.copy_(res)
    else: dst += res

  @staticmethod
  def ray_indices_u16(matrix, offset, B, D, H, W, h, w):
    """Contract of the saved index tensor (crn_ray_project / crn_ray_sample_fwd_idx): flat pixel iy*w+ix, 0xFFFF outside."""
    iy, ix, keep = O.ray_sample_indices(matrix.view(B, 4, 4).float(), offset.view(B, 3).float(), (D, H, W), (w, h))
    inside = keep & (iy >= 1) & (iy <= h) & (ix >= 1) & (ix <= w)
    return t.where(inside, (iy - 1) * w + (ix - 1), t.full_like(iy, 0xFFFF))

  def ray_project(self, matrix, offset, B, D, H, W, h, w, idx):
    assert h * w < 65535 and idx.dtype == t.int16       # (uint16 values in int16 storage: the cast wraps)
    idx.view(B, D, H, W).copy_(self.ray_indices_u16(matrix, offset, B, D, H, W, h, w).to(t.int32).to(t.int16))

  def ray_sample_fwd_idx(self, fmap, map_sB, B, Cn, h, w, matrix, offset, out, out_sB, D, H, W, idx, map_sC=None, map_sP=1):
    self.ray_sample_fwd(fmap, map_sB, B, Cn, h, w, matrix, offset, out, out_sB, D, H, W, map_sC=map_sC, map_sP=map_sP)
    self.ray_project(matrix, offset, B, D, H, W, h, w, idx)

  def ray_sample_bwd_idx(self, dout, dout_sB, B, Cn, D, H, W, idx, dmap, dmap_sB, h, w, zero_first=True):
    """index_put_(accumulate=True) from the saved indices alone (ray_traced_skip_connection.py:135, autograd)."""
    g = t.as_strided(dout, (B, Cn, D, H, W), (dout_sB, D * H * W, H * W, W, 1), dout.storage_offset())
    po = idx.view(B, D, H, W).to(t.int64) & 0xFFFF
    keep = po != 0xFFFF
    flat = t.zeros(B * h * w + 1, Cn, dtype=g.dtype)
    bb = t.arange(B)[:, None, None, None].expand_as(po)
    lin = t.where(keep, bb * (h * w) + po, t.full_like(po, B * h * w)).reshape(-1)
    flat.index_add_(0, lin, g.permute(0, 2, 3, 4, 1).reshape(-1, Cn))
    res = flat[:-1].view(B, h, w, Cn).permute(0, 3, 1, 2)
    dst = t.as_strided(dmap, (B, Cn, h, w), (dmap_sB, h * w, w, 1), dmap.storage_offset())
    if zero_first: dst.copy_(res)
    else: dst += res

  # -- losses / metrics / optimizer ---------------------------------------------------------
  LOSSES = {0: "iou_fgbg", 1: "xent_times_iou_agnostic", 2: "iou_agnostic", 3: "xent",
            4: "xent_times_iou_fgbg"}

  @t.enable_grad()
  def loss_fwd_bwd(self, kind, logits, gt_i32, B, Cn, S, loss, dlogits, grad_scale=1.0, weights=None):
    l = logits.detach().clone().requires_grad_(dlogits is not None)
    w = None if weights is None else weights.view(l.shape[0], *l.shape[2:])
    v = getattr(O, self.LOSSES[kind])(gt_i32.long().view(l.shape[0], *l.shape[2:]), l, w)
    loss.fill_(float(v))
    if dlogits is not None:
      v.backward()
      dlogits.copy_(l.grad * grad_scale)

  def softmax_superres(self, logits, m, B, Cn, D, H, W, out):
    # super_resolution.py:105-112 (reshape/permute) after the per-offset softmax (:124)
    pm = logits.view(m, m, m, B, Cn, D, H, W).softmax(dim=4)
    out.view(B, Cn, D * m, H * m, W * m).copy_(pm.permute(3, 4, 5, 0, 6, 1, 7, 2).reshape(B, Cn, m * D, m * H, m * W))

  def argmax_confusion(self, logits, gt_i32, B, Cn, S, labels, cm):
    lab = logits.argmax(1)
    if labels is not None: labels.copy_(lab.to(t.int32))
    if gt_i32 is not None:
      cm += O.confusion_matrix(gt_i32, lab, Cn).reshape(-1)

  def adam_step(self, p, g, m, v, n, lr, b1, b2, eps, grad_scale, step):
    gg = g * grad_scale
    m.add_((gg - m) * (1 - b1))
    v.mul_(b2).add_(gg * gg * (1 - b2))
    bc1 = 1 - b1 ** step; bc2 = 1 - b2 ** step
    p.sub_((lr / bc1) * (m / (v.sqrt() / np.sqrt(bc2) + eps)))

  def adam_set_hyper(self, hyper, lr, b1, b2, eps, grad_scale, step):
    hyper[:7] = t.tensor([lr, b1, b2, eps, grad_scale, 1 - b1 ** step, np.sqrt(1 - b2 ** step)], dtype=hyper.dtype)

  def adam_step_hyper(self, p, g, m, v, n, hyper):
    lr, b1, b2, eps, gs, bc1, bc2s = [float(x) for x in hyper[:7]]
    gg = g * gs
    m.add_((gg - m) * (1 - b1))
    v.mul_(b2).add_(gg * gg * (1 - b2))
    p.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))

  def transform_meshes(self, triangles, tri_mesh, mesh_matrix, out):
    m = mesh_matrix[tri_mesh.long()]                                  # [T,4,4]
    pts = t.cat([triangles, t.ones_like(triangles[..., :1])], -1)     # [T,3,4]
    r = t.einsum("tnm,tvm->tvn", m, pts)
    out.copy_(r[..., :3] / r[..., 3:4])

  def merge_labels(self, meshes_grid, scene_start, labels, B, D, H, W, sub_grid, out):
    """batched_example.py:186-196: out[b] = int32(max over the scene's meshes of label_m * grid_m); sub_grid: the
    grids are (2D+1)(2H+1)(2W+1) and only the odd centres are read (voxelization.py:167-182); no meshes -> 0."""
    g = meshes_grid[:, 1::2, 1::2, 1::2] if sub_grid else meshes_grid
    for b in range(B):
      lo, hi = int(scene_start[b]), int(scene_start[b + 1])
      if hi == lo:
        out[b].zero_()
      else:
        out[b].copy_((labels[lo:hi].to(g.dtype)[:, None, None, None] * g[lo:hi]).max(0).values.to(t.int32))

  def add_i64(self, p, n, v):
    p += v

  def zero(self, x):
    x.zero_()

  def fill_voxels(self, grid, out):
    out.copy_(t.as_tensor(O.fill_inside_voxels(grid.numpy())))
