set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-side 2>/dev/null | python tools/ms.py; done
timeout 300 python tools/phase_times.py 12 2>/dev/null | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q > $O/g15_tests.log 2>&1; tail -3 $O/g15_tests.log
