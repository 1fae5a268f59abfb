set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof3; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/phase_times.py 12 > $O/phase_times.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $R/tools/prof_step.py bf16x3 10 > /dev/null 2>&1
cp /tmp/prof/r03_kernel_stats.csv $O/step_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/prof/r03_kernel_trace.csv 13 > $O/step_trace_summary.txt
cp /tmp/prof/r03_kernel_trace.csv $O/step_kernel_trace.csv
