#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_optim.py tests/test_gpu_pretrain.py -x -q > gpurun_out/r4d_tests_dp.log 2>&1
echo "dp tests rc=$?" >> gpurun_out/r4d_tests_dp.log; tail -5 gpurun_out/r4d_tests_dp.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "bench_stack or ffn_skips or direct_parameter" > gpurun_out/r4d_tests_models.log 2>&1
echo "model tests rc=$?" >> gpurun_out/r4d_tests_models.log; tail -5 gpurun_out/r4d_tests_models.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt" > gpurun_out/r4d_tests_nt.log 2>&1
echo "nt tests rc=$?" >> gpurun_out/r4d_tests_nt.log; tail -3 gpurun_out/r4d_tests_nt.log
timeout 600 python bench.py > gpurun_out/r4d_bench.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r4d_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d[k] for k in ('value','ms_per_step')}, 'frac', d['roofline']['frac'], 'avg us', d['roofline']['avg_launch_us'])
    for o in d.get('other_configs',[]): print(o)
else:
    print(open('gpurun_out/r4d_bench.log').read()[-3000:])
PY
grep -h "bench stack\|bench-stack" gpurun_out/parity_report.txt | tail -5
