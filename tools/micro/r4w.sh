#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or attn" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/r4w_bench.json 2>/dev/null
python - <<'PY'
import json,os
j=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4w_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_us'], j['gemm_tn_roofline']['frac'])
for r in j['roofline_hbm'][:6]: print(r['kernel'], r['avg_launch_us'], r['frac'])
PY
