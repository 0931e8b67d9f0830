#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_mvit.py -x -q -m gpu > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3i_tests.log
tail -3 gpurun_out/r3i_tests.log
VTX_LIB=$PWD/videotransformer-pytorch_amd/libvtx_poolold.so timeout 300 python tools/micro/pool_bwd_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/r3i_pool_old.txt
timeout 300 python tools/micro/pool_bwd_bench.py 2>/dev/null | grep -v amdgpu > gpurun_out/r3i_pool_new.txt
paste gpurun_out/r3i_pool_old.txt gpurun_out/r3i_pool_new.txt
timeout 300 python tools/maskfeat_bench.py 32 3 > gpurun_out/r3i_maskfeat.txt 2>&1; tail -1 gpurun_out/r3i_maskfeat.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mfprof -- python $R/tools/maskfeat_bench.py 32 3 > /tmp/mf.log 2>&1
python $R/tools/rocpd_stats.py /tmp/mfprof > $R/gpurun_out/r3i_maskfeat_kernel_stats.csv
