# Regenerates the profiles/r06_* files in one gpurun call (every step under its own timeout):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh'   then copy gpurun_out/prof/r06_* to profiles/
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 3 --classes 14 --no-cpu-baseline > $O/r06_bench_m9.json 2>> $O/bench.err
timeout 300 python tools/layer_times.py 2 70 bf16x3 > $O/r06_layer_times_bf16x3.txt 2>/dev/null
timeout 300 python tools/layer_times.py 2 70 fp32 > $O/r06_layer_times_fp32.txt 2>/dev/null
timeout 300 python tools/layer_times.py 14 70 bf16x3 > $O/r06_layer_times_bf16x3_c14.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 > $O/r06_phase_times.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 14 > $O/r06_phase_times_c14.txt 2>/dev/null
timeout 900 bash tools/pmc_traffic.sh $O/r06_pmc_traffic.json bf16x3 > /dev/null 2>&1
timeout 900 bash tools/pmc_traffic.sh $O/r06_pmc_traffic_fp32.json fp32 > /dev/null 2>&1
timeout 900 bash tools/pmc_traffic.sh $O/r06_pmc_traffic_c14.json bf16x3 "--classes 14" > /dev/null 2>&1
timeout 300 bash tools/pmc_mfma.sh fwd s6c1 $O/r06_conv_mfma_pmc_bf16x3_fwd.txt bf16x3 > /dev/null 2>&1
timeout 300 bash tools/pmc_mfma.sh dgrad s6c1 $O/r06_conv_mfma_pmc_bf16x3_dgrad.txt bf16x3 > /dev/null 2>&1
timeout 300 bash tools/pmc_mfma.sh wgrad s6c1 $O/r06_conv_mfma_pmc_bf16x3_wgrad.txt bf16x3 > /dev/null 2>&1
timeout 300 bash tools/pmc_ct14.sh $O/r06_ct14_pmc.txt 14 > /dev/null 2>&1
timeout 300 bash tools/pmc_ct14.sh $O/r06_bf3_pmc_c2.txt 2 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python $R/tools/prof_step.py bf16x3 10 > /dev/null 2>&1
cp /tmp/prof/r06_kernel_stats.csv $O/r06_step_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/prof/r06_kernel_trace.csv 13 > $O/r06_step_trace_summary.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof14 -o r06 -- python $R/tools/prof_step.py bf16x3 10 14 > /dev/null 2>&1
cp /tmp/prof14/r06_kernel_stats.csv $O/r06_step_kernel_stats_c14.csv
python $R/tools/trace_summary.py /tmp/prof14/r06_kernel_trace.csv 13 > $O/r06_step_trace_summary_c14.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side --no-m9-side > $O/r06_bench_under_rocprof.json 2>/dev/null
cp /tmp/pb/b_kernel_stats.csv $O/r06_bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb32 -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side --no-m9-side --math fp32 > /dev/null 2>&1
cp /tmp/pb32/b_kernel_stats.csv $O/r06_bench_kernel_stats_fp32.csv
cd $R
timeout 200 python tools/bench_small.py > $O/r06_small_kernels.txt 2>/dev/null
timeout 200 python tools/bench_ray.py 2>/dev/null | grep -v amdgpu > $O/r06_ray_kernels.txt
# the ray-sample kernels beside MFMA-dense neighbours: the probe's failing family and the library's own launches
{ for m in 0 1 32 33 44 48; do timeout 120 python tools/mfma_neighbour.py probe $m 2>/dev/null | grep neighbour; done
  for l in "fwd s6c1" "fwd s6t1" "dgrad s6t1" "fwd s5t1" "dgrad s5t1" "dgrad s6c1" "wgrad s6c1"; do timeout 120 python tools/mfma_neighbour.py $l bf16x3 2>/dev/null | grep neighbour; done; } > $O/r06_mfma_neighbour.txt
timeout 600 python tools/project_glitch.py 20 2>/dev/null | grep neighbour > $O/r06_project_glitch.txt
timeout 200 python tools/train_synthetic.py 300 2 > $O/r06_train_synthetic.txt 2>/dev/null
timeout 200 python tools/run_noise.py 2 0 2>/dev/null | grep -v amdgpu > $O/r06_run_to_run_spread.txt
timeout 200 python tools/run_noise.py 4 30000 2>/dev/null | grep -v amdgpu >> $O/r06_run_to_run_spread.txt
timeout 200 python tools/cpu_enqueue.py bf16x3 2>/dev/null | grep graph= > $O/r06_graph_vs_eager.txt
DEBUG_HIP_FORCE_GRAPH_QUEUES=1 timeout 200 python tools/cpu_enqueue.py bf16x3 2>/dev/null | grep graph= | sed 's/^/DEBUG_HIP_FORCE_GRAPH_QUEUES=1: /' >> $O/r06_graph_vs_eager.txt
{ timeout 200 python tools/host_ahead.py 2 6 2>/dev/null | grep live; timeout 200 python tools/host_ahead.py 14 6 2>/dev/null | grep live; } > $O/r06_host_ahead.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -o v -- python $R/tools/bench_voxelize.py > $O/r06_voxelize.txt 2>/dev/null
cp /tmp/pv/v_kernel_stats.csv $O/r06_voxelize_kernel_stats.csv
cd $R
bash tools/trace_step.sh bf16x3 > /dev/null 2>&1; cp $R/gpurun_out/timeline.txt $O/r06_timeline.txt     # one step launch by launch (durations are real; gaps are stretched by the profiler)
ls -la $O
echo "copy gpurun_out/prof/r06_* to profiles/, commit, then: python tools/check_profiles_fresh.py"
