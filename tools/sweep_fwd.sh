for L in e2b e3b e4b e5b s5c1 s5t1 s4c1 s4t1 s3c1 s3t1 s6t1; do
 for mode in fwd dgrad; do
 for cfg in 8,1 8,2 4,1 4,2 4,4 2,1 2,2 2,4 1,1 1,2 1,4; do
  r=$(CRN_FWD_FORCE=$cfg CRN_DEBUG=1 python tools/bench_conv.py $mode $L 10 2>&1 | grep -E "TFLOP|crn_conv_fwd" | sort -u | tr "\n" " ")
  echo "$L $mode $cfg :: $(echo $r | grep -oE "MSUB [0-9]+ NSUB [0-9]+ CC [0-9]+ tile [0-9x]+ grid [0-9x]+ lds [0-9]+") :: $(echo $r | grep -oE "[0-9.]+ us +[0-9.]+ TFLOP")"
 done; done
done
