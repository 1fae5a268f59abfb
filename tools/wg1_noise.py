#!/usr/bin/env python
"""Run-to-run spread of crn_conv_wgrad_2d_bf3 on the rt_skip_5 compress conv (259 -> 12 channels, 64 x 64 map): the same call
N times, alone and next to a busy second stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend
from corenet_amd.model import conv_geometry as G
be = HipBackend()
for B, C, N, hw in ((2, 259, 12, 64), (4, 259, 12, 64), (2, 515, 24, 32), (2, 1027, 48, 16)):
  g = t.Generator().manual_seed(0)
  x = t.randn(B, C, hw, hw, generator=g).cuda(); dy = (t.randn(B, N, hw, hw, generator=g) * (t.rand(B, N, hw, hw, generator=g) < 0.2)).cuda()
  fwd = G.conv_fwd((N, C, 1, 1), 0)
  outs = []
  side = t.cuda.Stream()
  big = t.randn(4096, 4096, device="cuda")
  for mode in ("alone", "busy"):
    res = []
    for i in range(6):
      dw = t.zeros(C * fwd.npad, device="cuda")
      if mode == "busy":
        with t.cuda.stream(side):
          for _ in range(3): big @ big
      be.conv_wgrad(V.view_of(x), None, V.view_of(dy), dw, fwd.npad, fwd.window, fwd.pad_lo, False, math="bf16x3_2d")
      t.cuda.synchronize()
      res.append(dw.clone())
    ref = (x.double().flatten(2).transpose(0, 1).flatten(1) @ dy.double().flatten(2).transpose(0, 1).flatten(1).T)   # [C, N]
    got = res[0].view(C, fwd.npad)[:, :N].double()
    e_ref = float((got - ref).abs().max() / ref.abs().max())
    spread = max(float((r - res[0]).abs().max() / res[0].abs().max()) for r in res[1:])
    print(f"B={B} C={C} N={N} {hw}x{hw} {mode}: vs fp64 {e_ref:.1e}, run-to-run spread {spread:.1e}")
