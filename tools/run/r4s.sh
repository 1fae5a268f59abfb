python -m pytest tests/test_kernels_gpu.py -q 2>&1 | tail -3 > gpurun_out/r04_kern_t.log
python -m pytest tests/test_model_gpu.py -q 2>&1 | tail -3 > gpurun_out/r04_model_t.log
timeout 600 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'm9', d['m7_m9']['ms_per_step'], 'fp32', d['fp32_math']['ms_per_step'])" > gpurun_out/r04_bench_half.log 2>&1
