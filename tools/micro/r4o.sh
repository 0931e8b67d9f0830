#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 600 python bench.py > $O/r4o_bench.json 2>$O/r4o_bench.err; tail -c 600 $O/r4o_bench.err
python - <<'PY'
import json,os
j=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4o_bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['gemm_tn_roofline']['frac'])
for r in j['roofline_hbm']: print(r['kernel'], r['avg_launch_us'], r['frac'])
print([ (o['workload'][:30], o['clips_per_s']) for o in j['other_configs']])
PY
