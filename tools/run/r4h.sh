python -m pytest tests/test_kernels_gpu.py -q -k "ray" 2>&1 | tail -2 > gpurun_out/r04_ray_dbg.log
for tx in 16 8; do for d in 0 1 2; do echo "== TX4=$tx CRN_RAY_DBG=$d"; CRN_RAY_TX4=$tx CRN_RAY_DBG=$d python tools/bench_small.py 2>/dev/null | grep ray_sample_bwd | head -2; done; done >> gpurun_out/r04_ray_dbg.log 2>&1
