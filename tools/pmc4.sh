#!/bin/bash
# usage: pmc4.sh <mode> <layer> "<counters>"   (honours CRN_DBG_MODE etc. from the environment)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm4
rocprofv3 --kernel-trace --pmc $3 --output-format csv -d /tmp/pm4 -o a -- python $R/tools/bench_conv.py $1 $2 3 > /tmp/pm4.log 2>&1
grep -E "TFLOP" /tmp/pm4.log || tail -5 /tmp/pm4.log
python - <<PY
import csv,glob
for f in glob.glob("/tmp/pm4/*counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if "conv_" in r["Kernel_Name"]]
    agg={};n={}
    for r in rows:
        agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
    for k in sorted(agg): print("  %-28s %.4g"%(k, agg[k]/n[k]))
PY
