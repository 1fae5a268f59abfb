#!/usr/bin/env python
"""Wave-specialised bf16x3 kernel (slab path) against the round-2 kernel (fp32-weights path) on one layer: max |diff| and where.
usage: ws_diff.py <fwd|dgrad> <layer> [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend, Transform
from corenet_amd.model import conv_geometry as G
LAYERS = {"s6c1": ("conv", (16, 28, 5, 5, 5), 2, (64, 64, 64)), "s5c1": ("conv", (32, 56, 5, 5, 5), 2, (32, 32, 32)),
          "s6t1": ("convT", (16, 2, 7, 7, 7), 3, (64, 64, 64)), "s5t1": ("convT", (32, 16, 7, 7, 7), 3, (32, 32, 32)),
          "s4c1": ("conv", (64, 112, 5, 5, 5), 2, (16, 16, 16)), "t32": ("conv", (32, 56, 5, 5, 5), 2, (32, 32, 32))}
mode, key = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kind, wshape, pad, dims = LAYERS[key]
be = HipBackend(); g = t.Generator().manual_seed(0)
if kind == "conv":
  cin, cout = wshape[1], wshape[0]; fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad); odims = dims
else:
  cin, cout = wshape[0], wshape[1]; fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad); odims = tuple(2 * d for d in dims)
x = t.randn((B, cin) + dims, generator=g).cuda()
w = t.randn(wshape, generator=g) * 0.05
pk = lambda idx: t.where(t.as_tensor(idx) >= 0, w.reshape(-1)[t.as_tensor(idx).clamp(min=0).long()], t.zeros(())).cuda()
wf, wd = pk(fwd.index), pk(dgr.index)
tr = Transform((t.rand(cin) + 0.5).cuda(), t.randn(cin).cuda(), pre_relu=True)
nsf, nsd = G.slab_entries(fwd), G.slab_entries(dgr)
desc, blocks = G.operand_table([(0, 0, fwd, True), (wf.numel(), nsf, dgr, True)])
slabs = t.zeros((nsf + nsd) * 32, dtype=t.uint8, device="cuda")
be.bf3_operands(t.cat([wf, wd]), (t.as_tensor(desc).cuda(), blocks), slabs)
def run(slab):
  y = t.zeros((B, cout) + odims).cuda()
  yv = V.space_to_depth_view(V.view_of(y), (2, 2, 2), parity_major=True) if kind == "convT" else V.view_of(y)
  if mode == "fwd":
    be.conv_fwd(V.view_of(x), tr, wf, fwd.npad, None, 0, yv, fwd.window, fwd.pad_lo, 0, boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3", wslab=slabs[:nsf * 32] if slab else None)
    return y
  dy = t.randn((B, cout) + odims, generator=t.Generator().manual_seed(1)).cuda()
  dyv = V.space_to_depth_view(V.view_of(dy), (2, 2, 2), parity_major=True) if kind == "convT" else V.view_of(dy)
  dx = t.zeros_like(x)
  be.conv_fwd(dyv, None, wd, dgr.npad, None, 0, V.view_of(dx), dgr.window, dgr.pad_lo, 0, boxes=(dgr.n_boxes, dgr.c_boxes), math="bf16x3", wslab=slabs[nsf * 32:] if slab else None)
  return dx
a, b = run(False), run(True)
d = (a - b).abs()
print(f"{mode} {key} B={B}: max|old| {float(a.abs().max()):.3e}  max|diff| {float(d.max()):.3e}  differing {int((d > 0).sum())} of {d.numel()}")
if float(d.max()) > 0:
  idx = (d > 0).nonzero()
  for dim in range(1, 5):
    vals = idx[:, dim].unique()
    print(f"  dim {dim}: {len(vals)} distinct indices, first {vals[:12].tolist()}")
  b2 = run(True)
  print(f"  ws run-to-run max diff {float((b - b2).abs().max()):.3e}")
