for m in fwd dgrad; do
python tools/bench_conv.py $m s6c1 10 4 bf16x3 2>&1 | tail -1
CRN_DBG_MODE=1 python tools/bench_conv.py $m s6c1 10 4 bf16x3 2>&1 | tail -1
CRN_DBG_MODE=2 python tools/bench_conv.py $m s6c1 10 4 bf16x3 2>&1 | tail -1
CRN_DBG_MODE=3 python tools/bench_conv.py $m s6c1 10 4 bf16x3 2>&1 | tail -1
done
