#!/usr/bin/env python
"""Where did the side-stream ray scatter go wrong (round 4; the plan now runs it on the main stream)?  One training step, then the same scatter serially
on the step's own buffers: which pixels / channels / samples differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
B = 2
sd = O.make_state(0, 2, nbt=0)
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, 2)]
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.load_state_dict(sd); m.train()
for rep in range(3):
  m.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=0.0, adam_eps=1e-4)
  t.cuda.synchronize()
  p = m.engine.plan(B)
  got = p.gsmap[5].clone()
  p.gsmap[5].zero_()
  p._ray_bwd(5, p.dec[6]["gu"])
  t.cuda.synchronize()
  ref = p.gsmap[5].clone()
  d = (got - ref).abs()
  bad = d > 1e-5 * ref.abs().max()
  print(f"rep {rep}: {int(bad.sum())} of {bad.numel()} map elements differ; per sample {bad.sum((1, 2, 3)).tolist()}; per channel {bad.sum((0, 2, 3)).tolist()}")
  rows = bad.sum((0, 1, 3)); cols = bad.sum((0, 1, 2))
  print("   rows with differences:", [i for i in range(64) if rows[i] > 0])
  print("   cols with differences:", [i for i in range(64) if cols[i] > 0])
  sgn = (got - ref)[bad]
  if sgn.numel():
    print(f"   got - ref at those: min {float(sgn.min()):.3e} max {float(sgn.max()):.3e}; |ref| max {float(ref.abs().max()):.3e}; fraction with |got| < |ref|: {float((got[bad].abs() < ref[bad].abs()).float().mean()):.2f}")
