#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2; do for v in main lnnt; do
  if [ $v = main ]; then L=$R/videotransformer-pytorch_amd/libvtx.so; else L=$R/videotransformer-pytorch_amd/libvtx_$v.so; fi
  echo -n "$v: "; VTX_LIB=$L timeout 120 python tools/attn_bench.py 96 2>&1 | grep -E "layernorm" | tr '\n' ' '; echo
  echo -n "$v bench: "; VTX_LIB=$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done; done
