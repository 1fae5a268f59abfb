{
python tools/run_noise.py 2 0
echo "== CRN_ASYNC_SKIP=0"; CRN_ASYNC_SKIP=0 python tools/run_noise.py 2 0 | grep bf16x3
echo "== CRN_DEFER_REDUCE=0"; CRN_DEFER_REDUCE=0 python tools/run_noise.py 2 0 | grep bf16x3
echo "== CRN_BN_BWD_FUSE=0"; CRN_BN_BWD_FUSE=0 python tools/run_noise.py 2 0 | grep bf16x3
echo "== B=4 nbt=30000"; python tools/run_noise.py 4 30000
} > gpurun_out/r04_noise.log 2>&1
for w in 0 1; do CRN_BN_BWD_WIDE=$w timeout 600 python bench.py --no-cpu-baseline --no-fp32-side --no-m9-side --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('wide=$w ms_per_step', d['ms_per_step'])"; done > gpurun_out/r04_bnwide.log 2>&1
