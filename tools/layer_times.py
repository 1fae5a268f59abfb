#!/usr/bin/env python
"""Per-layer conv timings of one training step (HIP events around every conv launch, everything on one stream),
with the real FLOPs of every launch (2 x batch x logical output positions x real weights), TFLOP/s and the
fraction of the MFMA peak of the engine that ran it.
usage: layer_times.py [classes] [rows] [fp32|bf16x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
from corenet_amd.model.engine import BF16X3_LAUNCHES

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 4
MATH = sys.argv[3] if len(sys.argv) > 3 else "fp32"
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device="cuda", decoder_math=MATH)
m.load_state_dict(O.make_state(0, C)); m.train()
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, C)]
loss = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
for _ in range(2): m.train_step(image, v2s, off, grid, loss)
plan = m.engine.plan(B); plan.trace = []; plan.conv_positions = {}
t.cuda.synchronize()
a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
a.record(); m.train_step(image, v2s, off, grid, loss); b.record(); t.cuda.synchronize()
nreal = {k: int((cv.fwd.index >= 0).sum()) for k, cv in m.engine.convs.items()}
rows = []
for l, x, y in plan.trace:
  d, name = l.split()
  ms = x.elapsed_time(y)
  gf = 2.0 * B * plan.conv_positions[name] * nreal[name] / 1e9
  cvo = m.engine.convs[name]
  e2d = cvo.wop_kind == "e2d" and ((d == "fwd" and cvo.wop_f is not None) or (d == "dgrad" and cvo.wop_d is not None))
  bf3 = e2d or (MATH == "bf16x3" and (name, d) in BF16X3_LAUNCHES)
  peak = 2500.0 / 3 if bf3 else 157.3
  rows.append((ms, d, name, gf, gf / ms, gf / ms / peak, "e2d" if e2d else ("bf16x3" if bf3 else "fp32")))
tot = sum(r[0] for r in rows)
print(f"decoder_math={MATH} C={C} B={B}: step (serialized, with event overhead) {a.elapsed_time(b):.2f} ms; "
      f"{len(rows)} conv launches, {tot:.2f} ms, {sum(r[3] for r in rows):.0f} GFLOP")
print(f"{'us':>9} {'GFLOP':>8} {'TFLOP/s':>8} {'frac':>6} {'engine':>7}  launch      (frac: of 157.3 fp32 MFMA / of 2500/3 for bf16x3 and e2d = the encoder's split-bf16 engine;"
      " every launch is bracketed by HIP events on one stream: small launches include ~5 us of launch overhead)")
for ms, d, name, gf, tf, fr, eng in sorted(rows, reverse=True)[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
  print(f"{ms*1e3:9.1f} {gf:8.2f} {tf:8.1f} {fr:6.2f} {eng:>7}  {d:5s} {name}")
grp = {}
for ms, d, name, gf, tf, fr, eng in rows:
  k = d + ' ' + ('encoder' if 'encoder' in name else name.rsplit('.', 2)[0] if 'decoder' in name else 'other')
  g = grp.setdefault(k, [0.0, 0.0]); g[0] += ms; g[1] += gf
print("-- by stage")
for k in sorted(grp): print(f'{grp[k][0]*1e3:9.1f} us {grp[k][1]:8.2f} GFLOP {grp[k][1]/grp[k][0]:8.1f} TFLOP/s  {k}')
agg = {}
for ms, d, *_ in rows: agg[d] = agg.get(d, 0) + ms
print("-- by direction:", {k: round(v, 3) for k, v in agg.items()})
