#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streamed_one_phase" 2>&1 | tail -1
for i in 1 2 3; do
echo -n "prev: "; VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_prev.so timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "bwd spatial"
echo -n "new:  "; timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "bwd spatial"
done
