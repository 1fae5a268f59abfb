for rep in 1 2 3; do
  echo "per conv:            $(CRN_WG_BATCH=0 python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"
  echo "per block (default): $(python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"
done
python -m pytest tests/test_model_gpu.py tests/test_dist_gpu.py -x -q -m gpu -k "train_step or overlapped or decoder_gradients or deterministic or two_ranks or run_to_run or bench_two_rank" 2>&1 | grep -E "passed|failed|Error" | tail -3
