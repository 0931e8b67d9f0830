#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out
( for v in "" _abl1 _abl2 _abl3; do
  echo "== libvtx$v.so"
  VTX_LIB=$R/videotransformer-pytorch_amd/libvtx$v.so timeout 300 python tools/attn_bench.py 96 2>/dev/null | grep "bwd spatial"
done ) > gpurun_out/r3j_attn_fused_ablate.txt
cat gpurun_out/r3j_attn_fused_ablate.txt
