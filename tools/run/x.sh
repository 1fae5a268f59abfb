DET=1 python tools/run/fg.py 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/r04_fg.log
python tools/run/fg.py 2>&1 | grep -v amdgpu | tail -12 >> gpurun_out/r04_fg.log
