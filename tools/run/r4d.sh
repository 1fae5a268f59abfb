python -m pytest tests/test_kernels_gpu.py -q -k "fused_bn_bwd or bf16x3_fwd_dgrad" -s 2>&1 | grep -E "fused|ring|passed|failed|Error|error|assert" | tail -60 > gpurun_out/r04_bnf_t.log
python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r04_model_t.log
for f in 1 0; do
  CRN_BN_BWD_FUSE=$f timeout 600 python bench.py --no-cpu-baseline --no-fp32-side --steps 20 2> gpurun_out/r04_bench_f$f.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fuse=$f ms_per_step', d['ms_per_step'], 'm9', d['m7_m9']['ms_per_step'], 'loss', d['loss'], 'vox', d.get('roofline_voxelize'))"
done > gpurun_out/r04_bench_fuse.log 2>&1
