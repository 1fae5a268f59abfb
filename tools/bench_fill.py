#!/usr/bin/env python
"""fill_voxels micro-benchmark: 12 x 128^3 hollow shells."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as t
from corenet_amd.backend import HipBackend
be = HipBackend()
zz, yy, xx = np.meshgrid(*[np.arange(128)] * 3, indexing="ij")
d = np.sqrt((xx - 63.5) ** 2 + (yy - 63.5) ** 2 + (zz - 63.5) ** 2)
grid = t.tensor(np.stack([((d <= r) & (d > r - 1.5)).astype(np.float32) for r in (10, 30, 50)] * 4)).cuda()
out = t.empty_like(grid)
for _ in range(3): be.fill_voxels(grid, out)
t.cuda.synchronize(); a = t.cuda.Event(enable_timing=True); b = t.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): be.fill_voxels(grid, out)
b.record(); t.cuda.synchronize(); s = a.elapsed_time(b) / 10 * 1e-3
print(f"fill_voxels 12x128^3: {s*1e6:.1f} us  {8.0*grid.numel()/s/1e9:.0f} GB/s algorithmic ({8.0*grid.numel()/s/8e12*100:.1f}% of 8 TB/s)")
ref = t.tensor(np.stack([(d <= r).astype(np.float32) for r in (10, 30, 50)] * 4)).cuda()
print("exact:", bool((out == ref).all()))
