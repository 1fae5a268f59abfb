# forward / data-grad tile sweep: CRN_FWD_FORCE=MSUB,NSUB per layer (tools/bench_conv.py)
for L in ${LAYERS:-s6c1 s5c1 s4c1 s3c1 s6t1 s5t1 s4t1 s3t1 e2b e3b e4b e5b}; do
 for m in fwd dgrad; do
  echo "== $L $m default: $(CRN_DEBUG=1 python tools/bench_conv.py $m $L 5 2>&1 | grep -E "TFLOP|crn_conv_fwd" | sort -u | grep -oE "MSUB [0-9]+ NSUB [0-9]+ CC [0-9]+ tile [0-9x]+|[0-9.]+ us" | tr "\n" " ")"
  for cfg in 8,1 4,2 4,1 2,4 2,2 2,1 1,4 1,2; do
   r=$(CRN_FWD_FORCE=$cfg CRN_DEBUG=1 python tools/bench_conv.py $m $L 5 2>&1 | grep -E "TFLOP|crn_conv_fwd" | sort -u | grep -oE "MSUB [0-9]+ NSUB [0-9]+ CC [0-9]+ tile [0-9x]+ grid [0-9x]+|[0-9.]+ us" | tr "\n" " ")
   echo "   $cfg :: $r"
  done
 done
done
