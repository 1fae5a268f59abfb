set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 300 python bench.py --steps 20 --warmup 5 > $O/g6_bench.json 2> $O/g6_bench.err; head -c 1500 $O/g6_bench.json; echo
timeout 300 python tools/layer_times.py 2 70 bf16x3 > $O/g6_layer_times_bf16x3.txt 2>/dev/null
timeout 300 python tools/layer_times.py 14 70 bf16x3 > $O/g6_layer_times_bf16x3_c14.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 > $O/g6_phase_times.txt 2>/dev/null
timeout 300 python tools/phase_times.py 12 14 > $O/g6_phase_times_c14.txt 2>/dev/null
timeout 1800 python -m pytest tests -m gpu -x -q > $O/g6_full.log 2>&1; tail -5 $O/g6_full.log
