#!/usr/bin/env python
"""Only training steps (for rocprofv3 runs): prof_step.py [fp32|bf16x3] [steps] [classes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import bench
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
math = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
C = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device="cuda", decoder_math=math)
m.train()
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, C)]
grid = grid.to(t.int32)
loss = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
for _ in range(3): m.train_step(image, v2s, off, grid, loss)
t.cuda.synchronize()
a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps): m.train_step(image, v2s, off, grid, loss)
b.record(); t.cuda.synchronize()
print(f"{math} C={C}: {a.elapsed_time(b)/steps:.3f} ms/step")
