#!/usr/bin/env python
"""How long does the host take to ENQUEUE one training step (is the step GPU-bound or launch-bound)?
Launch by launch vs the captured HIP graph of CoreNet.train_step.  usage: cpu_enqueue.py [fp32|bf16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
math = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
for graph in (False, True):
  m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math=math)
  m.load_state_dict(O.make_state(0, 2)); m.train()
  for _ in range(3): loss = m.train_step(image, v2s, off, grid, "iou_fgbg", graph=graph)
  t.cuda.synchronize()
  n = 10
  t0 = time.perf_counter()
  for _ in range(n): loss = m.train_step(image, v2s, off, grid, "iou_fgbg", graph=graph)
  t1 = time.perf_counter()
  t.cuda.synchronize()
  t2 = time.perf_counter()
  print(f"{math} graph={graph}: enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step, loss {float(loss):.6f}")
  del m
