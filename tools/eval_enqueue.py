#!/usr/bin/env python
"""Eval-mode forward: host enqueue time vs total time per batch (is serving host-bound?), launch by launch and as a captured HIP
graph of the plan's forward (the graph path is an experiment of this tool, not of the product)."""
import os, sys, time
os.environ["CRN_EVAL_GRAPH"] = "0"          # the launch-by-launch leg below must be the launch-by-launch forward
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t, bench
from corenet_amd import _lib
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(B, 0, 2)]
m.eval()
with t.no_grad():
  for _ in range(3): m(image, v2s, off)
  t.cuda.synchronize()
  n = 20
  t0 = time.perf_counter()
  for _ in range(n): out = m(image, v2s, off)
  t1 = time.perf_counter(); t.cuda.synchronize(); t2 = time.perf_counter()
  print(f"eval forward B={B}, launch by launch: enqueue {1e3 * (t1 - t0) / n:.2f} ms, total {1e3 * (t2 - t0) / n:.2f} ms per batch")
  ref = out.clone()
  plan = m.engine.plan(B)
  plan.in_image.copy_(image); plan.in_v2s.copy_(v2s); plan.in_off.copy_(off)
  g = t.cuda.CUDAGraph(); cap = t.cuda.Stream()
  m.engine.be.splitk_reserve(cap)
  cap.wait_stream(t.cuda.current_stream())
  try:
    with t.cuda.graph(g, stream=cap), _lib.pinned_stream(cap):
      lg = plan.forward(plan.in_image, plan.in_v2s, plan.in_off, training=False)
    t.cuda.current_stream().wait_stream(cap)
    for _ in range(3): g.replay()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    t1 = time.perf_counter(); t.cuda.synchronize(); t2 = time.perf_counter()
    print(f"eval forward B={B}, graph replay: enqueue {1e3 * (t1 - t0) / n:.2f} ms, total {1e3 * (t2 - t0) / n:.2f} ms per batch; "
          f"equal to the launch-by-launch logits: {bool(t.equal(lg, ref))} (max diff {float((lg - ref).abs().max()):.1e})")
  except Exception as e:
    print("graph capture of the eval forward failed:", repr(e)[:300])
