#!/bin/bash
# Round 5, third GPU call: the rebuilt HOG kernel (selftest of all gradient pairs, bit-exact tests, rate), w4 tests, the bench line with TimeSformer-L/96.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -f $O/parity_report.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -9 > $O/r5c_smoke.txt; cat $O/r5c_smoke.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "hog or gemm_tn or selftest or maskfeat" > $O/r5c_tests.log 2>&1; echo "rc=$?" >> $O/r5c_tests.log
tail -6 $O/r5c_tests.log | cut -c1-300
timeout 300 python tools/other_configs.py hog 2>&1 | grep -v amdgpu.ids > $O/r5c_hog.txt; cat $O/r5c_hog.txt | cut -c1-400
timeout 600 python bench.py > $O/r5c_bench.log 2>&1; tail -1 $O/r5c_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline'])
for r in d.get('roofline_hbm', []): print('hbm', r.get('kernel', '')[:50], r.get('frac'), r.get('avg_launch_us'))
for r in d.get('other_configs', []): print('other', r.get('workload', '')[:70], r.get('clips_per_s'), r.get('ms_per_step'), r.get('skipped'), r.get('error'))
"
