#!/usr/bin/env python
"""The stem convolution (ZeroPad2d(3) + Conv2d 7x7 stride 2, 3 -> 64 on 256^2 images, resnet50.py:122-131) through the
C ABI the way the engine calls it (2x2 space-to-depth view of the image, 4x4 window over 12 channels).
usage: bench_stem.py [iters]     knobs: CRN_FWD_FORCE=M,N  CRN_FWD_SPLITS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend
from corenet_amd.model import conv_geometry as G
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
be = HipBackend()
g = G.stem_fwd((64, 3, 7, 7), 3)
w = t.randn(64, 3, 7, 7) * 0.05
idx = t.as_tensor(g.index)
wf = t.where(idx >= 0, w.reshape(-1)[idx.clamp(min=0).long()], t.zeros(())).cuda()
img = t.randn(4, 3, 256, 256).cuda(); y = t.zeros(4, 64, 128, 128).cuda()
xv = V.space_to_depth_view(V.view_of(img).channels(0, 3), (1, 2, 2), parity_major=False)
bias = t.randn(64).cuda()
run = lambda: be.conv_fwd(xv, None, wf, g.npad, bias, 0, V.view_of(y), g.window, g.pad_lo, 0, boxes=(g.n_boxes, g.c_boxes))
for _ in range(3): run()
t.cuda.synchronize(); a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): run()
b.record(); t.cuda.synchronize()
print(f"stem fwd {os.environ.get('CRN_FWD_FORCE')} splits {os.environ.get('CRN_FWD_SPLITS')}: {a.elapsed_time(b) / iters * 1e3:.1f} us")
# the stem's own kernels (csrc/stem_conv.hip)
def timeit(fn, label):
  for _ in range(3): fn()
  t.cuda.synchronize(); a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): fn()
  b.record(); t.cuda.synchronize()
  print(f"{label}: {a.elapsed_time(b) / iters * 1e3:.1f} us")
timeit(lambda: be.stem_conv_fwd(img, wf, bias, y, False), "stem_conv_fwd (no statistics)")
timeit(lambda: be.stem_conv_fwd(img, wf, bias, y, True), "stem_conv_fwd + BatchRenorm partial sums")
dy = t.randn(4, 64, 128, 128).cuda(); dw = t.zeros(wf.numel()).cuda()
timeit(lambda: be.conv_wgrad(xv, None, V.view_of(dy), dw, g.npad, g.window, g.pad_lo, False), "generic wgrad")
timeit(lambda: be.conv_wgrad(xv, None, V.view_of(dy), dw, g.npad, g.window, g.pad_lo, False, math="stem"),
       f"stem_conv_wgrad (CRN_STEM_WG_BLOCKS={os.environ.get('CRN_STEM_WG_BLOCKS')})")
