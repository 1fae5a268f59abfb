#!/usr/bin/env python
"""Elapsed time of the phases of one training step on the main stream (HIP events between the phases, no profiler):
together with the per-phase kernel time of a rocprofv3 trace this shows where the main stream idles.
usage: phase_times.py [steps] [classes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import bench
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
from corenet_amd.model.engine import LOSS_KINDS
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 2
LOSS = "iou_fgbg" if NC == 2 else "xent_times_iou_agnostic"
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), NC, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.train()
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, NC)]
grid = grid.to(t.int32)
print(f"C={NC} B=4 {LOSS}")
for _ in range(3): m.train_step(image, v2s, off, grid, LOSS)
eng = m.engine; plan = eng.plan(4)
plan.in_image.copy_(image); plan.in_v2s.copy_(v2s); plan.in_off.copy_(off); plan.gt.copy_(grid)
names = ["enc fwd", "dec fwd", "loss", "backward", "adam"]
ev = [[t.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] for _ in range(steps)]
t.cuda.synchronize()
joins = []
for s in range(steps):
  e = ev[s]
  eng.adam_step_graphable(4e-4 * 5, 1e-3, grad_scale=1.0, launch=False)
  eng.weights_dirty = True
  e[0].record()
  plan.forward_encoder(plan.in_image, True); e[1].record()
  plan.forward_decoder(plan.in_v2s, plan.in_off, True); e[2].record()
  eng.be.loss_fwd_bwd(LOSS_KINDS[LOSS], plan.logits, plan.gt, plan.B, eng.num_classes, 128 ** 3, plan.loss, plan.glogits, 1.0)
  e[3].record()
  orig_join = plan._join_side
  def timed_join():
    j0.record(); orig_join(); j1.record()
  j0, j1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  joins.append((j0, j1))
  plan._join_side = timed_join
  plan.backward(plan.glogits, grad_hook=eng.adam_bucket_hook()); e[4].record()
  plan._join_side = orig_join
  e[5].record()
t.cuda.synchronize()
tot = 0.0
for i, nm in enumerate(names):
  ms = sum(ev[s][i].elapsed_time(ev[s][i + 1]) for s in range(2, steps)) / (steps - 2)
  tot += ms
  print(f"{nm:10s} {ms:7.3f} ms")
print(f"main stream waits {sum(a.elapsed_time(b) for a, b in joins[2:]) / (steps - 2):.3f} ms for the side stream at the end of backward")
print(f"{'sum':10s} {tot:7.3f} ms; step to step {ev[2][0].elapsed_time(ev[steps - 1][0]) / (steps - 3):.3f} ms")
