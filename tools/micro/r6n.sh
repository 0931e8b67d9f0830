#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_mf
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_mf -- python $R/tools/maskfeat_bench.py 32 5 > /tmp/log_mf.txt 2>&1; tail -1 /tmp/log_mf.txt | cut -c1-200
python $R/tools/rocpd_stats.py /tmp/prof_mf > $O/r6n_maskfeat_kernel_stats.csv
