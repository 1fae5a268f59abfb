#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV of tools/prof_step.py: per-step kernel time by category, launches per
step, and how busy each HIP stream (queue) was.  usage: trace_summary.py <kernel_trace.csv> <steps incl. warm-up>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
# steady state only: from the 4th step's first kernel (crn_preprocess_caffe) to the last step's -- the first step also
# creates the plan (hundreds of torch zero-fills that round 2's summary counted as 49 "memset" launches per step)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pre = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
if len(pre) >= 6:
  rows = rows[pre[3]:pre[-1]]
  nsteps = len(pre) - 4
def cat(n):
  for key, c in (("conv_e2d", "encoder bf16x3 fwd/dgrad"), ("wgrad1x1_bf3", "encoder bf16x3 wgrad"), ("wgrad3x3_bf3", "encoder bf16x3 wgrad"),
                 ("bf3_operands", "weight pack / grad un-pack"), ("conv_bf3_wgrad", "conv bf16x3 wgrad"), ("conv_bf3", "conv bf16x3 fwd/dgrad"), ("conv_wgrad", "conv fp32 wgrad"),
                 ("conv_fwd", "conv fp32 fwd/dgrad"), ("pointwise", "conv fp32 1x1"), ("splitk", "split-K reduce"), ("bn_", "BatchRenorm"),
                 ("copy_tiles", "weight pack / grad un-pack"), ("copy_mats", "weight pack / grad un-pack"), ("ray_sample", "ray sample"), ("ray_scatter", "ray sample"), ("ray_project", "ray sample"), ("loss_", "loss"), ("adam", "adam"),
                 ("affine_add_relu", "residual add"), ("relu_bwd", "residual add"), ("maxpool", "stem pool"), ("Fill", "memset/zero"),
                 ("zero_", "memset/zero"), ("copyBuffer", "copies"), ("elementwise", "torch elementwise")):
    if key in n: return c
  return "other"
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
  a = agg[cat(r["Kernel_Name"])]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"{len(rows)} kernel launches over {nsteps} steps = {len(rows)/nsteps:.0f} per step; wall {((t1-t0)/1e6)/nsteps:.2f} ms per step")
tot = sum(v[1] for v in agg.values())
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
  print(f"{ms/nsteps:8.3f} ms/step {n/nsteps:7.1f} launches/step {100*ms/tot:5.1f}%  {k}")
print(f"{tot/nsteps:8.3f} ms/step  sum of kernel durations (streams overlap)")
q = collections.defaultdict(float)
for r in rows: q[r.get("Queue_Id", "?")] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, ms in sorted(q.items(), key=lambda kv: -kv[1]): print(f"queue {k}: busy {ms/nsteps:.3f} ms/step")
