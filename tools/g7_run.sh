set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
( time timeout 800 python bench.py ) > $O/g7_bench.json 2> $O/g7_bench.err; grep "bench.py \[" $O/g7_bench.err; tail -4 $O/g7_bench.err; head -c 600 $O/g7_bench.json
