python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "loss" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "train_step or decoder_gradients or deterministic or run_to_run or train_forward_backward or trajectory" 2>&1 | grep -E "passed|failed|Error" | tail -3
for rep in 1 2 3; do echo "step: $(python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"; done
