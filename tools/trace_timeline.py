#!/usr/bin/env python
"""Timeline of ONE steady-state training step out of a rocprofv3 --kernel-trace CSV of tools/prof_step.py: every launch
in start order with its queue, start offset, duration and the gap to the previous launch on the same queue.
usage: trace_timeline.py <kernel_trace.csv> [step index from the end, default 2] [max rows]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pre = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
step = rows[pre[-1 - back]:pre[-back]]
t0 = int(step[0]["Start_Timestamp"])
last_end = {}
def short(n):
  n = re.sub(r"\(anonymous namespace\)::|void |crnk::", "", n)
  return re.sub(r"\(.*", "", n)[:44]
qs = sorted({r["Queue_Id"] for r in step})
print(f"{len(step)} launches, {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
for r in step:
  s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
  gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
  last_end[q] = e
  print(f"{(s - t0) / 1e3:9.1f} {'  ' * qs.index(q)}q{qs.index(q)} {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short(r['Kernel_Name'])}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
