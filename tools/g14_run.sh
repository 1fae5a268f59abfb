set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
for v in 1 0 1 0; do echo "CRN_BIAS_GRAD_SIDE=$v"; CRN_BIAS_GRAD_SIDE=$v timeout 300 python bench.py --steps 30 --warmup 5 --classes 14 --no-cpu-baseline 2>/dev/null | python tools/ms.py; done
for v in 1 0 1 0; do echo "CRN_BIAS_GRAD_SIDE=$v"; CRN_BIAS_GRAD_SIDE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-side --no-m9-side 2>/dev/null | python tools/ms.py; done
