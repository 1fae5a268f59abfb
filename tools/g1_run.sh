set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "strided_views or fallback_branches or width_one or victims_beside or fill_known or fill_random or two_ranks_one_gpu" > $O/g1_tests.log 2>&1; tail -5 $O/g1_tests.log
timeout 300 python tools/layer_times.py 14 70 bf16x3 > $O/r06_layer_times_bf16x3_c14.txt 2>/dev/null
timeout 300 python tools/layer_times.py 2 70 bf16x3 > $O/r06_layer_times_bf16x3.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 14 > $O/r06_phase_times_c14.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 2 > $O/r06_phase_times.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof14 -o r06 -- python $R/tools/prof_step.py bf16x3 10 14 > $O/prof_step_c14.log 2>&1
cp /tmp/prof14/r06_kernel_stats.csv $O/r06_step_kernel_stats_c14.csv
python $R/tools/trace_summary.py /tmp/prof14/r06_kernel_trace.csv 13 > $O/r06_step_trace_summary_c14.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o r06 -- python $R/tools/prof_step.py bf16x3 10 2 > $O/prof_step_c2.log 2>&1
cp /tmp/prof2/r06_kernel_stats.csv $O/r06_step_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/prof2/r06_kernel_trace.csv 13 > $O/r06_step_trace_summary.txt
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/g1_bench.json 2> $O/g1_bench.err
tail -c 600 $O/g1_bench.json
