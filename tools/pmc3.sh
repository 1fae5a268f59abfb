#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pm3 -o a -- python $R/tools/bench_conv.py $1 $2 3 > /tmp/pm3.log 2>&1
grep -E "TFLOP" /tmp/pm3.log
python - <<PY
import csv,glob
for f in glob.glob("/tmp/pm3/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
    agg={};n={}
    for r in rows:
        agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
    for k in sorted(agg): print("  %-26s %.4g   per wave-chunk %.1f"%(k, agg[k]/n[k], agg[k]/n[k]/8192/7))
PY
