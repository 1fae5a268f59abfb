#!/bin/bash
# MFMA utilisation of one conv layer from hardware counters (own rocprofv3 --pmc pass, kernel-trace only):
#   usage (GPU box): bash tools/pmc_mfma.sh <fwd|dgrad|wgrad> <layer> [out.txt] [fp32|bf16x3]
# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (duration x SIMDs): the busy counter adds the cycles each SIMD's matrix pipe
# is occupied (32 per v_mfma_f32_16x16x4_f32); duration = GRBM_GUI_ACTIVE / 8 (the counter is summed over the XCDs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=${3:-gpurun_out/mfma_pmc.txt}; case "$OUT" in /*) ;; *) OUT=$R/$OUT ;; esac
MATH=${4:-fp32}
MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32; [ "$MATH" = bf16x3 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16
rm -rf /tmp/pm2
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES $MOPS SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm2 -o a -- python $R/tools/bench_conv.py $1 $2 5 4 $MATH > /tmp/pm2.log 2>&1
python - "$1" "$2" "$MATH" "$MOPS" > "$OUT" <<'PY'
import csv, glob, sys
MOPS = sys.argv[4]
print(f"rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES {MOPS} SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -- python tools/bench_conv.py {sys.argv[1]} {sys.argv[2]} 5 4 {sys.argv[3]}")
print("\n".join(l for l in open("/tmp/pm2.log").read().splitlines() if "TFLOP" in l))
for f in glob.glob("/tmp/pm2/*counter_collection.csv"):
  rows = [r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
  by = {}
  for r in rows:
    by.setdefault((r["Kernel_Name"].split("(")[0], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
  names = sorted({k[0] for k in by})
  for nm in names:
    g = {c: sum(v) / len(v) for (k, c), v in by.items() if k == nm}
    n = len(by[(nm, "GRBM_GUI_ACTIVE")])
    print(f"{nm}  ({n} launches, averages per launch)")
    for c, v in sorted(g.items()): print(f"  {c:32s} {v:.6g}")
    simds, xcds = 256 * 4, 8       # rocprofv3 adds GRBM_GUI_ACTIVE over the 8 XCDs
    clk = g['GRBM_GUI_ACTIVE'] / xcds
    print(f"  kernel duration = GUI_ACTIVE / {xcds} XCDs = {clk:.6g} shader clocks")
    print(f"  MfmaUtil = MFMA_BUSY / (duration x {simds} SIMDs) = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / (clk * simds) * 100:.1f} %")
    print(f"  MFMA_BUSY / (BUSY_CU_CYCLES x 4 SIMDs) = {g['SQ_VALU_MFMA_BUSY_CYCLES'] / (g['SQ_BUSY_CU_CYCLES'] * 4) * 100:.1f} %  (CUs idle at the ramp and tail excluded)")
    print(f"  FLOP from {MOPS} (512 flop per MOP) = {g[MOPS] * 512:.4g}" + ("  (3 bf16 MFMAs per fp32-equivalent product)" if "BF16" in MOPS else ""))
PY
cat "$OUT"
