#!/bin/bash
# Process-level A/B of library builds on the step: bash tools/micro/lib_ab.sh <rounds> libA.so libB.so ...   (names under videotransformer-pytorch_amd/)
R=$GRAFT_REPO_ROOT; cd $R; P=$R/videotransformer-pytorch_amd; N=$1; shift
for i in $(seq 1 $N); do for lib in "$@"; do
  VTX_LIB=$P/$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], 'nt', d['roofline']['avg_launch_us'], 'tn', d.get('gemm_tn_roofline',{}).get('ms_per_step'))"
done; done
