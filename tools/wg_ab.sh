# weight gradient of the bf16x3 engine, layer by layer: compiler-scheduled LDS reads (pipe 0) vs software-pipelined (pipe 1)
for l in s6c1 s5c1 s4c1 s3c1 s6t1 s5t1 s4t1 s3t1 s6t1c14; do for p in 0 1; do echo "pipe $p: $(CRN_BF3_WG_PIPE=$p python tools/bench_conv.py wgrad $l 20 4 bf16x3 2>&1 | tail -1)"; done; done
