# A/B of the wave-specialised bf16x3 forward / data-gradient kernel against the round-2 kernel, layer by layer
for l in s6c1 s5c1 s6t1 s5t1 s4c1 s4t1; do for m in fwd dgrad; do
  a=$(CRN_BF3_WS=0 python tools/bench_conv.py $m $l 20 4 bf16x3 2>&1 | tail -1)
  b=$(CRN_BF3_WS=1 python tools/bench_conv.py $m $l 20 4 bf16x3 2>&1 | tail -1)
  echo "old: $a"; echo "ws : $b"
done; done
