#!/bin/bash
# Round 6, second GPU call: the direct-from-L2 wprod kernel (tests, rate), the full-depth TimeSformer-L golden, the step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -f $O/parity_report.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wprod or merged or dropped" > $O/r6b_tests.log 2>&1; echo "rc=$?" >> $O/r6b_tests.log; tail -4 $O/r6b_tests.log
timeout 300 python tools/wprod_bench.py 2>&1 | grep -v amdgpu.ids > $O/r6b_wprod.txt; cat $O/r6b_wprod.txt
timeout 900 python -m pytest tests/test_gpu_00_baseline_configs.py -x -q -k "full_depth or t8_train or other" > $O/r6b_tests2.log 2>&1; echo "rc=$?" >> $O/r6b_tests2.log; tail -4 $O/r6b_tests2.log
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -k "merged or other_resolution or temporal" > $O/r6b_tests3.log 2>&1; echo "rc=$?" >> $O/r6b_tests3.log; tail -4 $O/r6b_tests3.log
grep -i "depth 24\|other resolution\|reference autocast output" $O/parity_report.txt | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --no-other-configs > $O/r6b_bench.log 2>&1; tail -1 $O/r6b_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
"
