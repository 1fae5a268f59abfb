#!/bin/bash
# usage: tools/pmc_conv.sh <mode> <layer>   -> prints averaged PMC counters of the conv kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pm1 -o a -- python $R/tools/bench_conv.py $1 $2 3 > /tmp/pm1.log 2>&1
grep -E "TFLOP" /tmp/pm1.log
python - <<PY
import csv,glob
for f in glob.glob("/tmp/pm1/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
    agg={};n={}
    for r in rows:
        agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
    for k in sorted(agg): print("  %-22s %.4g"%(k, agg[k]/n[k]))
    if rows: print("  VGPR", rows[0].get("VGPR_Count"), "scratch", rows[0].get("Scratch_Size"), "LDS", rows[0].get("LDS_Block_Size"))
PY
