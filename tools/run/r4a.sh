# round 4, first contact: whole GPU suite, ring kernel vs the launches it replaces, the bench line
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_gputest.log
for l in "fwd s6c1" "dgrad s6c1" "fwd s6t1" "dgrad s6t1" "fwd s5c1" "dgrad s5c1" "fwd s5t1" "dgrad s5t1" "fwd s6t1c14" "dgrad s6t1c14"; do
  CRN_RING_STAMPS=1 timeout 120 python tools/bench_conv.py $l 20 4 ring 2>&1 | tail -3
done > gpurun_out/r04_ring_b.log 2>&1
timeout 600 python bench.py > gpurun_out/r04_bench0.json 2> gpurun_out/r04_bench0.err
