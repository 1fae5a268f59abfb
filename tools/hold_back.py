#!/usr/bin/env python
"""Is the training step bound by the host's enqueue rate?  The GPU is held back with a spin kernel before each step, so that the host has
enqueued the whole step before the first kernel starts; the step's GPU time after the hold is compared with the free-running one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import bench
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.train()
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
for _ in range(5): m.train_step(image, v2s, off, grid, "iou_fgbg")
t.cuda.synchronize()
# calibrate _sleep
e = [t.cuda.Event(enable_timing=True) for _ in range(2)]
e[0].record(); t.cuda._sleep(10_000_000); e[1].record(); t.cuda.synchronize()
per_ms = 10_000_000 / e[0].elapsed_time(e[1])
print(f"_sleep: {per_ms:.0f} cycles per ms")
for delay_ms in (0.0, 2.0, 5.0, 8.0):
  res = []
  for rep in range(12):
    t.cuda.synchronize()
    a, b, c = [t.cuda.Event(enable_timing=True) for _ in range(3)]
    a.record()
    if delay_ms > 0: t.cuda._sleep(int(delay_ms * per_ms))
    b.record()
    t0 = time.perf_counter()
    m.train_step(image, v2s, off, grid, "iou_fgbg")
    t1 = time.perf_counter()
    c.record(); t.cuda.synchronize()
    res.append((b.elapsed_time(c), (t1 - t0) * 1e3))
  res = res[2:]
  print(f"GPU held back {delay_ms:.0f} ms before the step: step takes {sum(r[0] for r in res) / len(res):.3f} ms on the GPU after the hold (host enqueue {sum(r[1] for r in res) / len(res):.2f} ms)")
