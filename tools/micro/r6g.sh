#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "layernorm or colsum or reduce or gemm_tn" > $O/r6g_tests.log 2>&1; echo "rc=$?" >> $O/r6g_tests.log; tail -3 $O/r6g_tests.log
timeout 900 python -m pytest tests/test_gpu_00_baseline_configs.py -x -q -k "full_depth" > $O/r6g_tests2.log 2>&1; echo "rc=$?" >> $O/r6g_tests2.log; tail -3 $O/r6g_tests2.log; grep "depth 24" $O/parity_report.txt | cut -c1-330
timeout 900 python tools/cu_contention.py --holds 0,8,16,0 --tn-cus 256,240 2>&1 | grep -v amdgpu.ids > $O/r6g_cu_contention_tn_cus.txt; cat $O/r6g_cu_contention_tn_cus.txt
timeout 600 python tools/gemm_cg.py 150528 2>&1 | grep -v amdgpu.ids > $O/r6g_gemm_cg.txt; cat $O/r6g_gemm_cg.txt
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/r6g_bench.log 2>&1; tail -1 $O/r6g_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
for r in d.get('roofline_hbm', []): print('hbm', r.get('kernel', '')[:50], r.get('frac'), r.get('avg_launch_us'), r.get('ms_per_step'))
"
