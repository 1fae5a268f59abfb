python tools/wg1_noise.py > gpurun_out/r04_wg1.log 2>&1
