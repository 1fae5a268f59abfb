# ring kernel: where the barrier time comes from (per-wave stamps), ablations
python -m pytest tests/test_model_gpu.py -q -k "all_parameter_gradients" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r04_grad_t.log
{
for l in "fwd s6c1" "dgrad s6c1"; do
  echo "== $l"; CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -4
  for d in 1 5 6; do echo "-- CRN_DBG_MODE=$d"; CRN_DBG_MODE=$d CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -4 | head -3; done
done
echo "== fwd s6c1 CRN_RING_SLIDE=0"; CRN_RING_SLIDE=0 CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py fwd s6c1 20 4 ring 2>&1 | tail -4
echo "== fwd s6c1 CRN_RING_WGS=512"; CRN_RING_WGS=512 CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py fwd s6c1 20 4 ring 2>&1 | tail -4
} > gpurun_out/r04_ring_c.log 2>&1
