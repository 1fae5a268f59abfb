#!/usr/bin/env python
"""Idle gaps on the main HIP stream of a rocprofv3 --kernel-trace CSV of tools/prof_step.py (the trace under the
profiler runs slower than the real step -- launches are serialised more -- but the LARGE gaps show where the main
stream waits for the side stream or for the host).  usage: trace_gaps.py <kernel_trace.csv> [min_us] [steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
q = collections.Counter(r["Queue_Id"] for r in rows)
main = q.most_common(1)[0][0]
rs = sorted((r for r in rows if r["Queue_Id"] == main), key=lambda r: int(r["Start_Timestamp"]))
# one steady-state step: between the last two image pre-processing kernels (the first launch of a forward)
first = [i for i, r in enumerate(rs) if "preprocess" in r["Kernel_Name"]]
a, b = first[-2], first[-1]
step = rs[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3
print(f"main queue {main}: last step {len(step)} launches, span {(t1 - t0) / 1e3:.0f} us, busy {busy:.0f} us, idle {(t1 - t0) / 1e3 - busy:.0f} us")
gaps = []
for x, y in zip(step, step[1:]):
  g = (int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3
  gaps.append((g, x["Kernel_Name"][:60], y["Kernel_Name"][:60], (int(x["End_Timestamp"]) - t0) / 1e3))
small = sum(g for g, *_ in gaps if g < min_us)
print(f"gaps < {min_us} us: {small:.0f} us in total ({sum(1 for g, *_ in gaps if g < min_us)} gaps, mean {small / max(1, sum(1 for g, *_ in gaps if g < min_us)):.2f} us)")
for g, x, y, at in sorted(gaps, reverse=True):
  if g >= min_us: print(f"{g:8.1f} us at {at:8.0f} us  after {x}\n{'':24s}before {y}")
