#!/usr/bin/env python
"""Micro-benchmark of the encoder engine (crn_conv2d_bf3, csrc/conv_e2d.hip) on the encoder's layer shapes at batch 4:
HIP events around `iters` back-to-back calls (kernel + split-K reduction when one is used), forward form (fused
transform + bias).  usage: bench_e2d.py [iters]      tuning knobs: CRN_E2D_FILL, CRN_E2D_SPLITS"""
import os, sys
if os.environ.get("CRN_E2D_DBG"):
  os.environ.setdefault("CRN_TOOLS_LIB", "1")        # the stamp read-back exists only in the tools build of the library
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend, Transform
from corenet_amd.model import conv_geometry as G

CASES = [("s2_3x3", (64, 64, 3, 3), 64), ("s3_3x3", (128, 128, 3, 3), 32), ("s4_3x3", (256, 256, 3, 3), 16),
         ("s5_3x3", (512, 512, 3, 3), 8), ("s2_1x1_64_256", (256, 64, 1, 1), 64), ("s2_1x1_256_64", (64, 256, 1, 1), 64),
         ("s3_1x1_512_128", (128, 512, 1, 1), 32), ("s3_1x1_128_512", (512, 128, 1, 1), 32),
         ("s4_1x1_1024_256", (256, 1024, 1, 1), 16), ("s4_1x1_256_1024", (1024, 256, 1, 1), 16),
         ("s5_1x1_2048_512", (512, 2048, 1, 1), 8), ("s5_1x1_512_2048", (2048, 512, 1, 1), 8)]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 4
be = HipBackend()
g = t.Generator().manual_seed(0)
tot = 0.0
for name, wshape, hw in CASES:
  cout, cin, k = wshape[0], wshape[1], wshape[2]
  fwd = G.conv_fwd(wshape, k // 2)
  w = t.randn(wshape, generator=g) * 0.05
  idx = t.as_tensor(fwd.index)
  wf = t.where(idx >= 0, w.reshape(-1)[idx.clamp(min=0).long()], t.zeros(())).cuda()
  desc, blocks = G.operand_table([(0, 0, fwd)])
  wop = t.zeros(G.operand_entries(fwd) * 32, dtype=t.uint8, device="cuda")
  be.bf3_operands(wf, (t.as_tensor(desc).cuda(), blocks), wop)
  x = t.randn((B, cin, hw, hw), generator=g).cuda(); y = t.zeros((B, cout, hw, hw)).cuda()
  tr = Transform((t.rand(cin) + 0.5).cuda(), t.randn(cin).cuda(), post_relu=True)
  bias = t.randn(cout).cuda()
  run = lambda: be.conv2d_bf3(V.view_of(x), tr, wop, fwd.npad, bias, 0, V.view_of(y), fwd.window, fwd.pad_lo)
  for _ in range(5): run()
  t.cuda.synchronize(); a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): run()
  b.record(); t.cuda.synchronize()
  us = a.elapsed_time(b) / iters * 1e3
  tot += us
  flop = 2.0 * B * hw * hw * cin * cout * k * k
  print(f"{name:18s} {us:7.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
  if int(os.environ.get("CRN_E2D_DBG", "0")) & 16:      # shader-clock stamps of workgroup 0 (see conv_e2d.hip)
    import ctypes
    st = (ctypes.c_longlong * 32)()
    be.lib.cdll.crn_e2d_debug_stamps(st)
    print("   stamps (cycles since kernel entry):", {i: st[i] for i in (0, 1, 2, 3, 4, 5, 6, 7, 30, 31) if st[i] > 0})
print(f"sum {tot:.1f} us")
