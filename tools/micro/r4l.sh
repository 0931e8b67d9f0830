#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streamed_one_phase or one_pass_equals" 2>&1 | tail -15
for f in 1 2; do timeout 120 python tools/attn_bench.py 96 attn_fused=$f 2>&1 | grep -E "options|spatial"; done
for f in 1 2; do timeout 120 python tools/attn_bench.py 8 attn_fused=$f 2>&1 | grep -E "options|bwd spatial"; done
