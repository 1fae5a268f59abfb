# the encoder's 1x1 layers (forward and data gradient): LDS-staged pointwise kernel (CRN_PW2=0) vs operands straight from HBM
for l in e2a0 e2c e2a e3c e3a e4c e4a e5c e5a; do for m in fwd dgrad; do
  a=$(CRN_PW2=0 python tools/bench_conv.py $m $l 50 4 fp32 2>&1 | tail -1); b=$(CRN_PW2=1 python tools/bench_conv.py $m $l 50 4 fp32 2>&1 | tail -1)
  echo "old: $a"; echo "pw2: $b"
done; done
