#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mvit.py tests/test_gpu_pretrain.py tests/test_gpu_dp.py tests/test_gpu_trainer.py -x -q -m gpu > gpurun_out/r3p_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3p_tests.log
tail -4 gpurun_out/r3p_tests.log
timeout 300 python tools/maskfeat_bench.py 32 3 > gpurun_out/r3p_maskfeat.txt 2>&1; tail -1 gpurun_out/r3p_maskfeat.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf -- python $GRAFT_REPO_ROOT/tools/maskfeat_bench.py 32 3 > /tmp/mf.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/prof_mf > $GRAFT_REPO_ROOT/gpurun_out/r3p_maskfeat_kernel_stats.csv
