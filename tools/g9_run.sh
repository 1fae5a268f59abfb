set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
bash tools/trace_step.sh bf16x3; cp gpurun_out/timeline.txt $O/g9_timeline_c2.txt
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); python tools/trace_summary.py $f 5 > $O/g9_trace_summary_c2.txt; cat $O/g9_trace_summary_c2.txt
