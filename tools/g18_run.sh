set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r06_bench.json 2> $O/bench.err; python tools/ms.py < $O/r06_bench.json; grep -c "no HBM-traffic\|no record" $O/bench.err
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q > $O/g18_full.log 2>&1; tail -3 $O/g18_full.log
