#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streamed_one_phase" 2>&1 | grep -E "Error|assert|passed|failed" | head -12
echo "old fused:"; timeout 120 python tools/attn_bench.py 96 attn_fused=1 2>&1 | grep -E "bwd spatial"
echo "stream:"; timeout 120 python tools/attn_bench.py 96 attn_fused=2 2>&1 | grep -E "bwd spatial"
VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_trace.so timeout 120 python tools/attn_timeline.py 96 2>&1 | grep -v Warning | tail -45 | cut -c1-230
