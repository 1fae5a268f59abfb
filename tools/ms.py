#!/usr/bin/env python
"""stdin: bench.py's JSON line -> the step times in it (tools/g*_run.sh A/B loops)."""
import json, sys
for line in sys.stdin:
  if line.startswith("{"):
    d = json.loads(line)
    out = ["%.3f ms/step" % d["ms_per_step"]]
    if "m7_m9" in d: out.append("m7_m9 %.3f" % d["m7_m9"]["ms_per_step"])
    if "fp32_math" in d: out.append("fp32 %.3f" % d["fp32_math"]["ms_per_step"])
    print("  ".join(out))
