python -m pytest tests/test_kernels_gpu.py -q -k "ray" 2>&1 | tail -2 > gpurun_out/r04_ray2.log
python -m pytest tests/test_model_gpu.py -q -k "deterministic" 2>&1 | tail -2 >> gpurun_out/r04_ray2.log
python tools/bench_small.py 2>/dev/null | grep ray_sample_bwd >> gpurun_out/r04_ray2.log
for z in 8 16 32; do echo "ZSEG=$z"; CRN_RAY_ZSEG=$z python tools/bench_small.py 2>/dev/null | grep ray_sample_bwd; done >> gpurun_out/r04_ray2.log
