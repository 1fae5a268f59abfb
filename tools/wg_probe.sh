#!/bin/bash
# wgrad ablation on one layer: full / no-MFMA / no-staging, block-count sweep, PMC
L=${1:-s6c1}
R=$GRAFT_REPO_ROOT
CRN_DEBUG=1 python tools/bench_conv.py wgrad $L 10 2>&1 | grep -E "crn_conv_wgrad|TFLOP" | sort -u
echo "--- dbg1 (no MFMA)"; CRN_DBG_MODE=1 python tools/bench_conv.py wgrad $L 10 | grep TFLOP
echo "--- dbg2 (no staging)"; CRN_DBG_MODE=2 python tools/bench_conv.py wgrad $L 10 | grep TFLOP
for nb in 448 504 512 1000 1024 1536 2048; do echo "--- blocks $nb"; CRN_WG_BLOCKS=$nb python tools/bench_conv.py wgrad $L 10 | grep TFLOP; done
bash tools/pmc2.sh wgrad $L
bash tools/pmc3.sh wgrad $L
