set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "parity_walk or victims_beside or beside_mfma_neighbours" > $O/g4_tests.log 2>&1; grep -E "parity walk|passed|failed|Error|error|hostile|probe mode" $O/g4_tests.log | head -60
timeout 300 python tools/layer_times.py 14 6 bf16x3 2>/dev/null | head -10
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "m9 or gradients_every_element or golden" > $O/g4_tests2.log 2>&1; tail -3 $O/g4_tests2.log
timeout 300 python bench.py --steps 10 --warmup 3 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 400
