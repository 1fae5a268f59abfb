set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 300 python tools/host_ahead.py 2 3 2>/dev/null | grep live | tee $O/g8_host_ahead.txt
timeout 300 python tools/host_ahead.py 2 6 2>/dev/null | grep live | tee -a $O/g8_host_ahead.txt
