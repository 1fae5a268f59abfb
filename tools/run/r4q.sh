{
for d in 9 11 7 0; do echo "== CRN_DBG_MODE=$d flags 0"; CRN_DBG_MODE=$d CRN_RING_FLAGS=0 CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py fwd s6c1 20 4 ring 2>&1 | tail -4 | head -3 | cut -c1-330; done
} > gpurun_out/r04_ring_e.log 2>&1
