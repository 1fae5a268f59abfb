set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -s -k "parity_walk or victims_beside" > $O/g2_tests.log 2>&1; grep -E "parity walk|passed|failed|Error|error" $O/g2_tests.log | head -40
