python -m pytest tests/test_kernels_gpu.py -q -k "bf16x3_fwd_dgrad" -s 2>&1 | grep -E "ring|passed|failed|Error|error" | tail -30 > gpurun_out/r04_ring_t.log
{
for l in "fwd s6c1" "dgrad s6c1" "fwd s6t1" "dgrad s6t1" "fwd s5c1" "dgrad s5c1" "fwd s5t1" "dgrad s5t1" "fwd s4c1" "dgrad s4c1" "fwd s6t1c14" "dgrad s6t1c14"; do
  CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -4 | head -3
done
echo "== SLIDE2=0"
for l in "dgrad s6c1" "fwd s5c1" "fwd s5t1"; do
  CRN_RING_SLIDE2=0 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -1
done
} > gpurun_out/r04_ring_d.log 2>&1
