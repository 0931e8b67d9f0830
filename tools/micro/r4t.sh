#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do
echo -n "base:   "; timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "bwd temporal"
echo -n "reread: "; VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_rr.so timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "bwd temporal"
done
VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_rr.so timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" 2>&1 | tail -2
