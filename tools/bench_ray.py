#!/usr/bin/env python
"""Ray-sample kernels alone at the four decoder scales in the plan's layouts (B = 4): gather, gather + saved index tensor,
projection alone, scatter from the saved indices, plain scatter (projection launch + scatter).  The gradient is the tail
channels of the stage's concat-buffer gradient (batch stride (cout + skip) * S), the map gradient is zeroed by the caller.
Tuning aids of the scatter: CRN_RAY_CN (4 | 12), CRN_RAY_ZSEG (8 | 16), CRN_RAY_TX (32 | 64); with the tools build
(CRN_TOOLS_LIB=1) CRN_RAY_DBG ablations: 1 no LDS adds, 2 no window write-out, 4 every plane = the segment's first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("CRN_RAY_DBG"):
  os.environ.setdefault("CRN_TOOLS_LIB", "1")
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.backend import HipBackend
be = HipBackend(); B = 4
def timeit(fn, n=20):
  for _ in range(3): fn()
  t.cuda.synchronize(); a = t.cuda.Event(enable_timing=True); b = t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(n): fn()
  b.record(); t.cuda.synchronize(); return a.elapsed_time(b) / n * 1e-3
for res, C, cout in ((64, 12, 16), (32, 24, 32), (16, 48, 64), (8, 96, 128)):
  S = res ** 3
  cmap_cl = t.randn(B, res, res, C).cuda()
  m = (O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3))[None].expand(B, 4, 4).reshape(B, 16).contiguous().cuda()
  off = t.full((B, 3), 0.5).cuda()
  u = t.zeros(B, cout + C, res, res, res).cuda(); gu = t.randn(B, cout + C, res, res, res).cuda()
  dmap = t.zeros(B, C, res, res).cuda(); idx = t.zeros(B, S, dtype=t.int16).cuda()
  by = 4.0 * B * C * (S + res * res)
  line = [f"{res}^3 x{C}:"]
  s = timeit(lambda: be.ray_sample_fwd(cmap_cl, cmap_cl.stride(0), B, C, res, res, m, off, u[:, cout:], u.stride(0), res, res, res, map_sC=1, map_sP=C))
  line.append(f"gather {s*1e6:.1f} us ({by/s/8e12:.3f})")
  s = timeit(lambda: be.ray_sample_fwd_idx(cmap_cl, cmap_cl.stride(0), B, C, res, res, m, off, u[:, cout:], u.stride(0), res, res, res, idx, map_sC=1, map_sP=C))
  line.append(f"gather+idx {s*1e6:.1f} us ({(by + 2.0 * B * S)/s/8e12:.3f})")
  s = timeit(lambda: be.ray_project(m, off, B, res, res, res, res, res, idx))
  line.append(f"project {s*1e6:.1f} us")
  s = timeit(lambda: be.ray_sample_bwd_idx(gu[:, cout:], gu.stride(0), B, C, res, res, res, idx, dmap, dmap.stride(0), res, res, False))
  line.append(f"scatter(idx) {s*1e6:.1f} us ({(by + 2.0 * B * S)/s/8e12:.3f} of 8 TB/s)")
  s = timeit(lambda: be.ray_sample_bwd(gu[:, cout:], gu.stride(0), B, C, res, res, res, m, off, dmap, dmap.stride(0), res, res, False))
  line.append(f"scatter(plain) {s*1e6:.1f} us")
  print("  ".join(line))
