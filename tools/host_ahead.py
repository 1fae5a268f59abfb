#!/usr/bin/env python
"""Is the step's GPU time bound by how fast the host enqueues it?  The same train steps timed twice with HIP events: (a) back to back
as bench.py runs them (the host enqueues while the GPU executes), (b) behind a spin kernel long enough that the host has enqueued all
of the timed steps before the GPU starts the first (pure GPU time of a fully pre-enqueued step).  usage: host_ahead.py [classes] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(4, 0, C)]
grid = grid.to(t.int32)
loss_name = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.reset_parameters(seed=0); m.train()
step = lambda: m.train_step(image, v2s, off, grid, loss_name, lr=4e-4, adam_eps=1e-4)
for _ in range(5): step()
t.cuda.synchronize()
ea, eb = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
for rep in range(3):
  ea.record()
  for _ in range(n): step()
  eb.record(); t.cuda.synchronize()
  live = ea.elapsed_time(eb) / n
  spin_ms = 9.0 * n + 10.0
  t0 = time.perf_counter()
  t.cuda._sleep(int(spin_ms * 1e-3 * 2.1e9))          # (cycles of the shader clock; ~2.1 GHz under load)
  ea.record()
  for _ in range(n): step()
  eb.record()
  enq = (time.perf_counter() - t0) * 1e3
  t.cuda.synchronize()
  ahead = ea.elapsed_time(eb) / n
  print(f"C={C}: live {live:.3f} ms/step; pre-enqueued behind a {spin_ms:.0f} ms spin (host enqueued {n} steps in {enq:.1f} ms) {ahead:.3f} ms/step")
