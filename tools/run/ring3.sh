python -m pytest tests/test_kernels_gpu.py -q -k "bf16x3_fwd_dgrad" -s 2>&1 | grep -E "ring|passed|failed|Error|error" | tail -40 > gpurun_out/r04_ring_t.log
python -m pytest tests/test_model_gpu.py -q -k "all_parameter_gradients" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r04_grad_t.log
