# ablations of the wave-specialised kernel: 1 = no MFMA (producer chain alone), 4 = no LDS commits, 5 = no global loads
for l in s6c1 s5c1; do for m in fwd dgrad; do for d in 0 1 4 5; do
  echo "dbg=$d $(CRN_DBG_MODE=$d python tools/bench_conv.py $m $l 20 4 bf16x3 2>&1 | tail -1)"
done; done; done
