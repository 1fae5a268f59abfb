#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm2
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES --output-format csv -d /tmp/pm2 -o a -- python $R/tools/bench_conv.py $1 $2 3 > /tmp/pm2.log 2>&1
grep -E "TFLOP" /tmp/pm2.log
python - <<PY
import csv,glob
for f in glob.glob("/tmp/pm2/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
    agg={};n={}
    for r in rows:
        agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
    for k in sorted(agg): print("  %-26s %.4g"%(k, agg[k]/n[k]))
for f in glob.glob("/tmp/pm2/*kernel_trace.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
    d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
    print("  duration ns avg", sum(d)/len(d))
PY
