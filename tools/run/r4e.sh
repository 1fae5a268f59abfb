for f in 1 0; do
echo "== CRN_BN_BWD_FUSE=$f"
CRN_BN_BWD_FUSE=$f python -m pytest tests/test_model_gpu.py -x -q -k "train_step_reduces_loss" -s 2>&1 | grep -E "autograd path|assert|Error|passed|failed|^E " | head -30
done > gpurun_out/r04_ts.log 2>&1
python -m pytest tests/test_kernels_gpu.py -q -k "fused_bn_bwd" 2>&1 | tail -3 >> gpurun_out/r04_ts.log
for q in 0 1 2 4 8; do echo "== DEBUG_HIP_FORCE_GRAPH_QUEUES=$q"; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q python tools/cpu_enqueue.py bf16x3 2>&1 | tail -2; done > gpurun_out/r04_graphq.log 2>&1
