#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_ex
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ex -- python $R/bench.py --stream fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > /tmp/log_ex.txt 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_ex > $O/r6l_kernel_stats_exact_b96.csv; head -3 $O/r6l_kernel_stats_exact_b96.csv | cut -c1-100
