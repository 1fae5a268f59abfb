#!/usr/bin/env python
"""Per-layer conv timings of one training step (HIP events around every conv launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 4
MATH = sys.argv[3] if len(sys.argv) > 3 else "fp32"
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device="cuda", decoder_math=MATH)
m.load_state_dict(O.make_state(0, C)); m.train()
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, C)]
loss = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
for _ in range(2): m.train_step(image, v2s, off, grid, loss)
plan = m.engine.plan(B); plan.trace = []
t.cuda.synchronize()
a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
a.record(); m.train_step(image, v2s, off, grid, loss); b.record(); t.cuda.synchronize()
rows = sorted(((x.elapsed_time(y), l) for l, x, y in plan.trace), reverse=True)
tot = sum(r[0] for r in rows)
print(f"step {a.elapsed_time(b):.2f} ms; conv launches {len(rows)} total {tot:.2f} ms")
geom = {}
for k, cv in m.engine.convs.items(): geom[k] = cv
for ms, l in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
  print(f"{ms*1e3:9.1f} us  {l}")
agg = {}
for ms, l in rows: agg[l.split()[0]] = agg.get(l.split()[0], 0) + ms
print(agg)
grp = {}
for ms, l in rows:
  k = l.split()[0] + ' ' + ('encoder' if 'encoder' in l else l.split()[1].rsplit('.', 2)[0] if 'decoder' in l else 'other')
  grp[k] = grp.get(k, 0) + ms
for k in sorted(grp): print(f'{grp[k]*1e3:9.1f} us  {k}')
