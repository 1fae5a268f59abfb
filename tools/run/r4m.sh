{ echo "== CRN_RAY_NOWIN=1"; CRN_RAY_NOWIN=1 python tools/run_noise.py 2 0 2>&1 | grep -A12 "^fp32.*rep 1" | grep "bf16x3\|gsmap\[5\]"; } > gpurun_out/r04_noise2.log 2>&1
