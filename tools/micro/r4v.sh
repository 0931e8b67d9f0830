#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
L=$R/videotransformer-pytorch_amd/libvtx_hf.so
for i in 1 2; do
for n in 1 2 4; do echo -n "head-fast attn_hw_bwd=$n: "; VTX_LIB=$L timeout 120 python tools/attn_bench.py 96 attn_hw_bwd=$n 2>&1 | grep -E "bwd temporal"; done
for n in 2 3 4 6 12; do echo -n "head-fast attn_hw_fwd=$n: "; VTX_LIB=$L timeout 120 python tools/attn_bench.py 96 attn_hw_fwd=$n 2>&1 | grep -E "fwd temporal"; done
done
echo -n "base fwd: "; timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "fwd temporal"
