{ python tools/run_noise.py 2 0 2>&1 | grep "^bf16x3\|^fp32"; python tools/run_noise.py 4 30000 2>&1 | grep "^bf16x3\|^fp32"; } | cut -c1-330 > gpurun_out/r04_noise3.log 2>&1
python -m pytest tests/test_model_gpu.py -q -k "graph_replay or distributed_data_parallel or train_step_reduces or all_parameter" 2>&1 | tail -4 > gpurun_out/r04_model_t.log
timeout 600 python bench.py --no-cpu-baseline --no-fp32-side --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'm9', d['m7_m9']['ms_per_step'])" > gpurun_out/r04_bench_raymain.log 2>&1
