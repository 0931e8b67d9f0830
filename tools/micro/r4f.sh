#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r4f_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/r4f_gpu_tests.log; tail -6 gpurun_out/r4f_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r4f_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4f_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline']['frac'], 'avg us', d['roofline']['avg_launch_us'], 'tn', d['gemm_tn_roofline'])
    for o in d.get('roofline_hbm',[]): print(o['kernel'], o['avg_launch_us'], o['frac'], o['ms_per_step'])
else:
    print(open('gpurun_out/r4f_bench.log').read()[-3000:])
PY
