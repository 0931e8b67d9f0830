#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/r4g_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $O/r4g_kernel_stats.csv
head -45 $O/r4g_kernel_stats.csv | cut -c1-220
