set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd $R
timeout 600 python tools/project_glitch.py 20 > $O/g3_project_glitch.txt 2>&1; cat $O/g3_project_glitch.txt | grep -v amdgpu
timeout 300 python tools/layer_times.py 14 12 bf16x3 2>/dev/null | head -20
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "bf16x3_eval_b4 or golden or m9" > $O/g3_tests.log 2>&1; tail -5 $O/g3_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 --classes 14 --no-cpu-baseline 2>/dev/null | head -c 600
