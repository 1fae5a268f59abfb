set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "parity_walk" > $O/g10_tests.log 2>&1; tail -3 $O/g10_tests.log
for st in 24 0 40; do echo "CRN_CT_WG_STAGE=$st"; CRN_CT_WG_STAGE=$st timeout 300 python tools/layer_times.py 14 4 bf16x3 2>/dev/null | grep "stage_6.t1"; done
echo EVEN; CRN_CT_WG_EVEN=1 timeout 300 python tools/layer_times.py 14 4 bf16x3 2>/dev/null | grep "wgrad decoder.stage_6.t1"
timeout 300 python bench.py --steps 20 --warmup 5 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 300; echo
CRN_CT_WG_EVEN=1 timeout 300 python bench.py --steps 20 --warmup 5 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 300; echo
