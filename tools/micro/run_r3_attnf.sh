#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" > gpurun_out/r3j_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3j_tests.log
tail -5 gpurun_out/r3j_tests.log
( echo "== two kernels (attn_fused=0)"; timeout 300 python tools/attn_bench.py 96 attn_fused=0 2>/dev/null | grep -i "attn"
  echo "== one pass (attn_fused=1)";  timeout 300 python tools/attn_bench.py 96 attn_fused=1 2>/dev/null | grep -i "attn" ) > gpurun_out/r3j_attn_fused_ab.txt
cat gpurun_out/r3j_attn_fused_ab.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | cut -c1-200
VTX_ATTN_FUSED=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown 2>/dev/null | tail -1 | cut -c1-200
