#!/usr/bin/env python
"""Run-to-run spread of the gradient slab of one training step (default mode: atomics arrive in any order), per math mode:
the same step from the same state on two model instances, max |g1 - g2| / max |g| over the slab and per gradient bucket.
usage: run_noise.py [B] [nbt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nbt = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sd = O.make_state(0, 2, nbt=nbt)
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, 2)]
gi = grid.to(t.int32)
for math in ("fp32", "bf16x3"):
  ms = []
  for _ in range(2):
    m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math=math)
    m.load_state_dict(sd); m.train(); ms.append(m)
  for rep in range(2):
    ls = [float(m.train_step(image, v2s, off, gi, "iou_fgbg", lr=0.0, adam_eps=1e-4)) for m in ms]
    t.cuda.synchronize()
    g1, g2 = ms[0].engine.store.grads, ms[1].engine.store.grads
    tot = float((g1 - g2).abs().max() / g1.abs().max())
    per = []
    for label, lo, hi in ms[0].engine.grad_buckets:
      a, b = g1[lo:hi], g2[lo:hi]
      per.append(f"{label or 'stem..stage3'} {float((a - b).abs().max() / a.abs().max().clamp(min=1e-30)):.1e}")
    worst = []
    lo3 = [lo for lb, lo, hi in ms[0].engine.grad_buckets if lb == "decoder.stage_3."][0]
    gdec = float(g1[lo3:].abs().max())
    for name, p1 in ms[0].named_parameters():
      a, b = ms[0].engine.store.view(name, grad=True), ms[1].engine.store.view(name, grad=True)
      if not name.startswith("decoder."): continue
      worst.append((float((a - b).abs().max() / gdec), name))
    worst.sort(reverse=True)
    print("   noisiest tensors: " + ", ".join(f"{n} {e:.1e}" for e, n in worst[:8]))
    if worst and worst[0][0] > 1e-5:
      nm = worst[0][1]
      a, b = ms[0].engine.store.view(nm, grad=True), ms[1].engine.store.view(nm, grad=True)
      d = (a - b).abs().flatten()
      top = d.topk(min(12, d.numel())).indices.tolist()
      shp = tuple(a.shape)
      import numpy as np
      print(f"   {nm} {shp}: {int((d > 1e-6 * gdec).sum())} of {d.numel()} elements differ by more than 1e-6 of the bucket's max; largest at "
            + ", ".join(str(tuple(int(v) for v in np.unravel_index(i, shp))) + f" {float(a.flatten()[i]):.3e}/{float(b.flatten()[i]):.3e}" for i in top))
    pa, pb = ms[0].engine.plan(B), ms[1].engine.plan(B)
    for k in (5, 4):
      ga, gb_ = pa.gsmap[k], pb.gsmap[k]
      dch = (ga - gb_).abs().amax((0, 2, 3)) / ga.abs().max()
      gu_a, gu_b = pa.dec[k + 1]["gu"], pb.dec[k + 1]["gu"]
      print(f"   gsmap[{k}] final per-channel diff: " + " ".join(f"{float(v):.0e}" for v in dch) +
            f" | gu[{k + 1}] diff {float((gu_a - gu_b).abs().max() / gu_a.abs().max()):.1e}")
    lg = float((ms[0].engine.plan(B).logits - ms[1].engine.plan(B).logits).abs().max())
    print(f"{math} B={B} nbt={nbt} rep {rep}: losses {ls[0]:.7f} {ls[1]:.7f}, logits max diff {lg:.1e}, slab {tot:.1e} | " + ", ".join(per))
  del ms
