set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "parity_walk" > $O/g11_tests.log 2>&1; tail -3 $O/g11_tests.log
for x in 1 0; do echo "CRN_CT_XIMG=$x"; CRN_CT_XIMG=$x timeout 300 python tools/layer_times.py 14 4 bf16x3 2>/dev/null | grep "stage_6.t1"; done
for x in 1 0 1 0; do CRN_CT_XIMG=$x timeout 300 python bench.py --steps 20 --warmup 5 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 240; echo; done
