#!/bin/bash
# Round 6: what the three launches of decoder stage_6.t1 at 14 classes (csrc/convt_par.hip) spend their cycles on -- matrix pipe, LDS,
# waiting -- from hardware counters, each group in its own rocprofv3 --pmc pass (kernel-trace only) over three m7 / m9 training steps.
#   usage (GPU box): bash tools/pmc_ct14.sh [out.txt] [classes, default 14; 2 = the h7 step: every split-bf16 decoder launch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/prof/r06_ct14_pmc.txt}; NC=${2:-14}; export NC; case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
rm -rf /tmp/pc1 /tmp/pc2 /tmp/pc3
mkdir -p "$(dirname "$OUT")"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc1 -o a -- python $R/tools/prof_step.py bf16x3 3 $NC > /tmp/pc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/pc2 -o a -- python $R/tools/prof_step.py bf16x3 3 $NC > /tmp/pc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/pc3 -o a -- python $R/tools/prof_step.py bf16x3 3 $NC > /tmp/pc3.log 2>&1
python - > "$OUT" <<'PY'
import csv, glob
by = {}
for d in ("/tmp/pc1", "/tmp/pc2", "/tmp/pc3"):
  for f in glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
      nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
      if "convt_" in nm or "conv_bf3" in nm:
        by.setdefault((nm, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
import os
print(f"three passes of: rocprofv3 --kernel-trace --pmc <group> -- python tools/prof_step.py bf16x3 3 {os.environ.get('NC', '14')}   (training steps; averages per launch over the launches of a kernel instance)")
for nm in sorted({k[0] for k in by}):
  g = {c: sum(v) / len(v) for (k, c), v in by.items() if k == nm}
  print(f"{nm}  ({len(by.get((nm, 'GRBM_GUI_ACTIVE'), []))} launches)")
  for c, v in sorted(g.items()): print(f"  {c:32s} {v:.6g}")
  if "GRBM_GUI_ACTIVE" in g:
    clk = g["GRBM_GUI_ACTIVE"] / 8
    print(f"  duration = GUI_ACTIVE / 8 XCDs = {clk:.6g} shader clocks; MfmaUtil = MFMA_BUSY / (duration x 1024 SIMDs) = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / (clk * 1024) * 100:.1f} %")
    if "SQ_LDS_IDX_ACTIVE" in g:
      print(f"  LDS array busy = LDS_IDX_ACTIVE / (duration x 256 CUs) = {g['SQ_LDS_IDX_ACTIVE'] / (clk * 256) * 100:.1f} %; bank-conflict cycles / LDS cycles = {g.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, g['SQ_LDS_IDX_ACTIVE']) * 100:.1f} %")
    if "SQ_WAVE_CYCLES" in g:
      w = g["SQ_WAVE_CYCLES"]
      print(f"  of a wave's cycles: parked (s_waitcnt / barrier) {g.get('SQ_WAIT_ANY', 0) / w * 100:.1f} %, issue-stalled {g.get('SQ_WAIT_INST_ANY', 0) / w * 100:.1f} % (LDS issue {g.get('SQ_WAIT_INST_LDS', 0) / w * 100:.1f} %), issuing {g.get('SQ_ACTIVE_INST_ANY', 0) / w * 100:.1f} %")
PY
cat "$OUT"
for f in /tmp/pc1.log /tmp/pc2.log /tmp/pc3.log; do grep -v "simple_timer\|output_stream" $f | head -n 12 | cut -c1-240; done
