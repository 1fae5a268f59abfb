for d in 0 1 2 3; do echo "== CRN_RAY_DBG=$d"; CRN_RAY_DBG=$d python tools/bench_small.py 2>/dev/null | grep ray_sample_bwd; done > gpurun_out/r04_ray_dbg.log 2>&1
