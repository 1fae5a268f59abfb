# sweep of tile configurations / split-K for the encoder 3x3 layers (tuning aid)
for l in e2b e3b e4b e5b; do for m in fwd dgrad; do
  echo "== $l $m default: $(python tools/bench_conv.py $m $l 20 4 2>&1 | tail -1)"
  for f in 8,1 4,2 4,1 2,4 2,2 2,1 1,4 1,2; do for sp in 1 2 4 8; do
    echo "$l $m force $f splits $sp: $(CRN_FWD_FORCE=$f CRN_FWD_SPLITS=$sp python tools/bench_conv.py $m $l 20 4 2>&1 | tail -1 | sed 's/.*: //')"
  done; done
done; done
