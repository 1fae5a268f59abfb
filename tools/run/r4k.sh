python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r04_gputest.log
timeout 600 python bench.py > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err
