#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r4s_tests.log 2>&1; tail -3 $O/r4s_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/r4s_bench.json 2>/dev/null
python - <<'PY'
import json,os
j=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4s_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['avg_launch_us'], j['gemm_tn_roofline']['frac'])
for r in j['roofline_hbm'][:6]: print(r['kernel'], r['avg_launch_us'], r['frac'])
PY
