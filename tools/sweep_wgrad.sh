for L in s5t1 s5c1 s6c1 s6t1; do
 for cfg in 4,8,8,1 4,4,8,1 2,4,8,1 4,2,8,1 4,4,8,2 4,2,8,2 2,2,8,2 4,4,4,2 2,4,4,2 4,2,4,2 2,4,4,4 2,2,4,4 4,2,4,4 4,4,2,4 2,4,2,4 4,8,4,1 4,4,4,1; do
  r=$(CRN_WG_FORCE=$cfg CRN_DEBUG=1 python tools/bench_conv.py wgrad $L 5 2>&1 | grep -E "TFLOP|crn_conv_wgrad" | sort -u | tr "\n" " ")
  echo "$L $cfg :: $(echo $r | grep -oE "RSUB [0-9]+ NSUB [0-9]+ CC [0-9]+ tile [0-9x]+ grid [0-9x]+ lds [0-9]+") :: $(echo $r | grep -oE "[0-9.]+ us +[0-9.]+ TFLOP")"
 done
done
