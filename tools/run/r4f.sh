python -m pytest tests/test_model_gpu.py -x -q -k "train_step_reduces_loss" -s 2>&1 | grep -E "autograd path|assert|Error|passed|failed|^E " | head -30 > gpurun_out/r04_ts.log
python -m pytest tests/test_kernels_gpu.py -q -k "ray" 2>&1 | tail -5 > gpurun_out/r04_ray_t.log
python tools/bench_small.py 2>/dev/null | grep ray_sample > gpurun_out/r04_ray_b.log
CRN_RAY_BWD2=0 python tools/bench_small.py 2>/dev/null | grep ray_sample_bwd | sed 's/^/old: /' >> gpurun_out/r04_ray_b.log
python -m pytest tests/test_model_gpu.py -x -q --deselect "tests/test_model_gpu.py::test_train_step_reduces_loss_and_matches_autograd_path" -k "not gradients" 2>&1 | tail -5 > gpurun_out/r04_model_t.log
