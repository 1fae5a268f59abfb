set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 3 > $O/r02_bench.json 2> $O/bench.err
python bench.py --steps 10 --warmup 3 --classes 14 --no-cpu-baseline > $O/r02_bench_m9.json 2>> $O/bench.err
python tools/layer_times.py 2 70 bf16x3 > $O/r02_layer_times_bf16x3.txt 2>/dev/null
python tools/layer_times.py 2 70 fp32 > $O/r02_layer_times_fp32.txt 2>/dev/null
python tools/phase_times.py 12 > $O/r02_phase_times.txt 2>/dev/null
bash tools/pmc_traffic.sh $O/r02_pmc_traffic.json bf16x3 > /dev/null 2>&1
bash tools/pmc_mfma.sh fwd s6c1 $O/r02_conv_mfma_pmc_bf16x3_fwd.txt bf16x3 > /dev/null 2>&1
bash tools/pmc_mfma.sh dgrad s6c1 $O/r02_conv_mfma_pmc_bf16x3_dgrad.txt bf16x3 > /dev/null 2>&1
bash tools/pmc_mfma.sh wgrad s6c1 $O/r02_conv_mfma_pmc_bf16x3_wgrad.txt bf16x3 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r02 -- python $R/tools/prof_step.py bf16x3 10 > /dev/null 2>&1
cp /tmp/prof/r02_kernel_stats.csv $O/r02_step_kernel_stats.csv
python $R/tools/trace_summary.py /tmp/prof/r02_kernel_trace.csv 13 > $O/r02_step_trace_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-side > $O/r02_bench_under_rocprof.json 2>/dev/null
cp /tmp/pb/b_kernel_stats.csv $O/r02_bench_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -o e -- python $R/tools/bench_e2d.py 20 > /dev/null 2>&1
python $R/tools/bench_e2d_trace.py /tmp/pe/e_kernel_trace.csv 20 > $O/r02_e2d_kernel_times.txt
cd $R
python tools/cpu_enqueue.py > $O/r02_graph_vs_eager.txt 2>/dev/null
(for k in 3x3 all; do for st in "" 2 23 45 345 2345; do CRN_E2D_KINDS=$k CRN_E2D_FWD_STAGES=$st python tools/e2d_parity.py 2>/dev/null; done; done; CRN_E2D=0 python tools/e2d_parity.py 2>/dev/null) > $O/r02_e2d_parity.txt
