#!/bin/bash
# HBM traffic of the two roofline kernels of bench.py, per launch (MI355X_MICROARCH.md recipe):
# separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, kernel-trace only; FETCH_SIZE x2 on gfx950.
# usage (on the GPU box): bash tools/pmc_traffic.sh <out.json> [fp32|bf16x3]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-$R/gpurun_out/pmc_traffic.json}
MATH=${2:-bf16x3}
EXTRA=${3:-}          # extra bench.py arguments, e.g. "--classes 14" (profiles/r06_pmc_traffic_c14.json)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pt_$c -o a -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-side --no-m9-side --math $MATH $EXTRA > /tmp/pt_$c.log 2>&1
done
python - "$OUT" "$MATH $EXTRA" <<'PY'
import csv, glob, json, sys
def collect(c):
  acc, seq = {}, {}
  for f in glob.glob(f"/tmp/pt_{c}/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
      nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
      key = (nm.split("(")[0].strip(), int(r["Grid_Size"]))
      a = acc.setdefault(key, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
      seq.setdefault(key, []).append(float(r["Counter_Value"]))
  return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}, seq
F, nF, sF = collect("FETCH_SIZE"); W, nW, sW = collect("WRITE_SIZE")
out = {"command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-side --math {sys.argv[2]}",
       "note": "KB per launch averaged over the launches of that (kernel, grid size), per_launch_hbm_bytes in dispatch order (4 steps: 1 warm-up + 3); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request); separate passes for the two counters",
       "kernels": {}}
for k in sorted(F, key=lambda k: -F[k] - W.get(k, 0)):
  name, grid = k
  if not any(s in name for s in ("conv_fwd_kernel", "conv_wgrad_kernel", "conv_bf3", "convt_", "ray_sample", "ray_scatter", "fill_fused", "pointwise")): continue
  f_kb = 2.0 * F[k]; w_kb = W.get(k, 0.0)
  out["kernels"][f"{name} grid {grid}"] = {"launches": nF[k], "FETCH_SIZE_KB_x2": round(f_kb, 1), "WRITE_SIZE_KB": round(w_kb, 1),
                                           "hbm_bytes": int((f_kb + w_kb) * 1024)}
  # several layers can share (kernel, grid): keep the per-launch values in dispatch order as well
  if nF[k] <= 64 and k in sW and len(sW[k]) == len(sF[k]):
    out["kernels"][f"{name} grid {grid}"]["per_launch_hbm_bytes"] = [int((2.0 * a + b) * 1024) for a, b in zip(sF[k], sW[k])]
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in list(out["kernels"].items())[:14]: print(k, v)
PY
