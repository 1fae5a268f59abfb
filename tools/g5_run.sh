set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "parity_walk or resident_weights" > $O/g5_tests.log 2>&1; grep -E "parity walk cout 14 B|resident weights|passed|failed|Error|error" $O/g5_tests.log | head -40
timeout 300 python tools/layer_times.py 14 4 bf16x3 2>/dev/null | head -8
timeout 300 python tools/layer_times.py 2 14 bf16x3 2>/dev/null | head -18
timeout 300 python bench.py --steps 10 --warmup 3 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 300; echo
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-side --no-m9-side 2>/dev/null | head -c 300; echo
timeout 1500 python -m pytest tests -m gpu -x -q > $O/g5_full.log 2>&1; tail -5 $O/g5_full.log
