timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "bf16x3_fwd_dgrad or fused_bn" -s 2>&1 | grep -E "conv3d_k5_64|passed|failed|Error|error" | tail -12 > gpurun_out/r04_half_t.log
{
for h in 1 0; do echo "== CRN_BF3_HALF=$h"; CRN_BF3_HALF=$h timeout 120 python tools/bench_conv.py fwd s6c1 30 4 bf16x3 2>&1 | tail -2; done
} > gpurun_out/r04_half_b.log 2>&1
