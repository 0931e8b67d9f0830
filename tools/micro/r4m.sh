#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streamed_one_phase" 2>&1 | tail -5
echo "old fused:"; timeout 120 python tools/attn_bench.py 96 attn_fused=1 2>&1 | grep -E "bwd spatial"
echo "full:"; timeout 120 python tools/attn_bench.py 96 attn_fused=2 2>&1 | grep -E "bwd spatial"
for v in $(ls videotransformer-pytorch_amd/libvtx_sa*.so | sed 's/.*_sa//; s/.so//'); do echo "ablate $v:"; VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_sa$v.so timeout 120 python tools/attn_bench.py 96 attn_fused=2 2>&1 | grep -E "bwd spatial"; done
