python tools/layer_times.py 14 40 bf16x3 2>&1 | grep -v amdgpu > gpurun_out/r04_layers_m9.txt
