set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 300 python tools/phase_times.py 12 2>/dev/null
timeout 300 python tools/phase_times.py 12 2>/dev/null
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-side --no-m9-side 2>/dev/null | python tools/ms.py
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -k "rccl or dist or exchange or bucket or replicas or two_rank or native" > $O/g16_tests.log 2>&1; tail -3 $O/g16_tests.log
