python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04_gputest.log
