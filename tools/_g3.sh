python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "decoder_inputs" 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "golden or eval_b4 or other_image" 2>&1 | grep -E "passed|failed|Error" | tail -3
for rep in 1 2 3; do echo "step: $(python tools/prof_step.py bf16x3 30 2>/dev/null | grep ms/step)"; done
