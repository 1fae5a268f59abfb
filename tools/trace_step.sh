#!/bin/bash
# One training step's launch-by-launch timeline (tools/trace_timeline.py) -> gpurun_out/timeline.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/tools/prof_step.py ${1:-bf16x3} 8 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out
python $R/tools/trace_timeline.py $f > $R/gpurun_out/timeline.txt
head -3 $R/gpurun_out/timeline.txt
