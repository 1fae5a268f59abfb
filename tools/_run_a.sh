set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "bf16x3 or ray" 2>&1 | tail -8 ) > gpurun_out/a_tests.txt
for rs in 0 1; do
  echo "== CRN_BF3_ROWSKIP=$rs" >> gpurun_out/a_bench.txt
  CRN_BF3_ROWSKIP=$rs bash tools/bf3bench.sh bf16x3 "fwd dgrad" "s5t1 s4t1 s3t1 s6t1 s6t1c14" >> gpurun_out/a_bench.txt 2>&1
done
cat gpurun_out/a_tests.txt gpurun_out/a_bench.txt
