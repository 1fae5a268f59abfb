cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "stem" 2>&1 | tail -15 ) > gpurun_out/b_tests.txt
cat gpurun_out/b_tests.txt
( python tools/bench_stem.py 50; CRN_STEM_WG_BLOCKS=512 python tools/bench_stem.py 50 | tail -1; CRN_STEM_WG_BLOCKS=128 python tools/bench_stem.py 50 | tail -1 ) > gpurun_out/b_bench.txt 2>&1
cat gpurun_out/b_bench.txt
( timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/b_model.txt
cat gpurun_out/b_model.txt
( python bench.py --steps 20 --warmup 5; CRN_STEM=0 python bench.py --steps 20 --warmup 5 ) > gpurun_out/b_bench_step.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/b_bench_step.txt'):
  if l.startswith('{'):
    d=json.loads(l); print('ms_per_step', d['ms_per_step'], 'm7_m9', d.get('m7_m9',{}).get('ms_per_step'), 'fp32', d.get('fp32_math',{}).get('ms_per_step'))
PY
