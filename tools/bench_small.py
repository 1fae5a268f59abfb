#!/usr/bin/env python
"""Micro-benchmarks of the HBM-bound kernels: ray-sample fwd/bwd at the four decoder scales,
fill_voxels at 12 x 128^3, fused loss, BatchRenorm at the largest activation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as t
from oracle import corenet_oracle as O
from corenet_amd.backend import HipBackend
be = HipBackend(); B = 4
def timeit(fn, n=20):
  for _ in range(3): fn()
  t.cuda.synchronize(); a = t.cuda.Event(enable_timing=True); b = t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); t.cuda.synchronize(); return a.elapsed_time(b) / n * 1e-3
for res, C in ((64, 12), (32, 24), (16, 48), (8, 96)):
  cmap = t.randn(B, C, res, res).cuda()
  cmap_cl = cmap.permute(0, 2, 3, 1).contiguous()      # the model's layout: channel-last skip map
  m = (O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3))[None].expand(B, 4, 4).reshape(B, 16).contiguous().cuda()
  off = t.full((B, 3), 0.5).cuda()
  out = t.zeros(B, C, res, res, res).cuda(); g = t.randn(B, C, res, res, res).cuda(); dmap = t.zeros_like(cmap)
  s = timeit(lambda: be.ray_sample_fwd(cmap_cl, C * res * res, B, C, res, res, m, off, out, C * res ** 3, res, res, res, map_sC=1, map_sP=C))
  by = 4.0 * B * C * (res ** 3 + res * res)
  print(f"ray_sample_fwd {res}^3 x{C}: {s*1e6:.1f} us  {by/s/1e9:.0f} GB/s ({by/s/8e12*100:.0f}% of 8 TB/s)")
  s = timeit(lambda: be.ray_sample_bwd(g, C * res ** 3, B, C, res, res, res, m, off, dmap, C * res * res, res, res, True))
  print(f"ray_sample_bwd {res}^3 x{C}: {s*1e6:.1f} us  {by/s/1e9:.0f} GB/s")
zz, yy, xx = np.meshgrid(*[np.arange(128)] * 3, indexing="ij")
d = np.sqrt((xx - 63.5) ** 2 + (yy - 63.5) ** 2 + (zz - 63.5) ** 2)
grid = t.tensor(np.stack([((d <= r) & (d > r - 1.5)).astype(np.float32) for r in (10, 30, 50)] * 4)).cuda()
out = t.empty_like(grid)
s = timeit(lambda: be.fill_voxels(grid, out), 10)
print(f"fill_voxels 12x128^3: {s*1e6:.1f} us  {8.0*grid.numel()/s/1e9:.0f} GB/s algorithmic ({8.0*grid.numel()/s/8e12*100:.1f}% of 8 TB/s)")
logits = t.randn(B, 2, 128, 128, 128).cuda(); gt = t.randint(0, 2, (B, 128, 128, 128), dtype=t.int32).cuda()
loss = t.zeros(1).cuda(); dl = t.zeros_like(logits)
s = timeit(lambda: be.loss_fwd_bwd(0, logits, gt, B, 2, 128 ** 3, loss, dl, 1.0))
by = logits.numel() * 4 * 3 + gt.numel() * 4 * 2
print(f"loss iou_fgbg fwd+bwd C=2: {s*1e6:.1f} us  {by/s/1e9:.0f} GB/s")
