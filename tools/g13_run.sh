set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s -k "parity_walk" > $O/g13_tests.log 2>&1; grep -E "wgrad \(|passed|failed|rror" $O/g13_tests.log | head -20
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "m9 or gradients_every_element or golden or graph or deterministic" > $O/g13_tests2.log 2>&1; tail -3 $O/g13_tests2.log
timeout 300 python tools/layer_times.py 14 4 bf16x3 2>/dev/null | grep "stage_6.t1"
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 240; echo; done
timeout 300 python tools/phase_times.py 12 14 2>/dev/null
