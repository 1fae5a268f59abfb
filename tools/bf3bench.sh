M=${1:-bf16x3}; MODES=${2:-"fwd dgrad wgrad"}; LAYERS=${3:-"s6c1 s5c1 s4c1 s3c1 s6t1 s5t1 s4t1 s3t1 s6t1c14"}
for l in $LAYERS; do for m in $MODES; do python tools/bench_conv.py $m $l 10 4 $M 2>&1 | tail -1; done; done
