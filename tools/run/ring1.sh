python -m pytest tests/test_kernels_gpu.py -q -k "bf16x3_fwd_dgrad" -x -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r04_ring_t.log
for l in "fwd s6c1" "dgrad s6c1" "fwd s6t1" "dgrad s6t1" "fwd s5c1" "dgrad s5c1" "fwd s5t1" "dgrad s5t1" "fwd s6t1c14" "dgrad s6t1c14"; do
  timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -2
done > gpurun_out/r04_ring_b.log 2>&1
python -m pytest tests/test_model_gpu.py -q -k "all_parameter_gradients" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r04_grad_t.log
