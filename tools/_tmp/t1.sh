cd $GRAFT_REPO_ROOT
CRN_BF3_STAMPS=1 timeout 100 python tools/bench_conv.py dgrad s6t1c14 5 4 bf16x3 2>&1 | tail -28
CRN_BF3_STAMPS=1 CRN_DEBUG=1 timeout 100 python tools/bench_conv.py dgrad s6t1c14 1 4 bf16x3 2>&1 | grep "crn_conv_fwd_bf3" | head -2
CRN_BF3_STAMPS=1 timeout 100 python tools/bench_conv.py fwd s6t1c14 5 4 bf16x3 2>&1 | tail -12
