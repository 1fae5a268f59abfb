#!/usr/bin/env python
"""How long does the host take to ENQUEUE one training step (is the step GPU-bound or launch-bound)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda")
m.load_state_dict(O.make_state(0, 2)); m.train()
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
for _ in range(3): m.train_step(image, v2s, off, grid, "iou_fgbg")
t.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): m.train_step(image, v2s, off, grid, "iou_fgbg")
t1 = time.perf_counter()
t.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
