set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "resident_weights or parity_walk" > $O/g17_tests.log 2>&1; tail -2 $O/g17_tests.log
for v in 1 0; do echo "CRN_CT_RES_XCD=$v"; CRN_CT_RES_XCD=$v timeout 300 python tools/layer_times.py 2 14 bf16x3 2>/dev/null | grep "stage_6.t1"; done
for v in 1 0 1 0; do echo "CRN_CT_RES_XCD=$v"; CRN_CT_RES_XCD=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-fp32-side --no-m9-side 2>/dev/null | python tools/ms.py; done
