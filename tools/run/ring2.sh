for l in "fwd s6c1" "fwd s6t1" "dgrad s6t1"; do
  CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -3 | head -2
done > gpurun_out/r04_ring_b2.log 2>&1
