#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "streamed_one_phase" 2>&1 | tail -1
for i in 1 2; do
for v in kv2 main kv4; do
  if [ $v = main ]; then L=$R/videotransformer-pytorch_amd/libvtx.so; else L=$R/videotransformer-pytorch_amd/libvtx_$v.so; fi
  echo -n "$v: "; VTX_LIB=$L timeout 120 python tools/attn_bench.py 96 attn_fused=2 2>&1 | grep -E "bwd spatial"
done; done
echo -n "old fused: "; timeout 120 python tools/attn_bench.py 96 attn_fused=1 2>&1 | grep -E "bwd spatial"
