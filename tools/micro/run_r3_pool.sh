#!/bin/bash
# round 3: pooling-convolution backward kernels, old build vs new on one box + the MViT tests
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mvit.py -x -q -m gpu > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3i_tests.log
tail -3 gpurun_out/r3i_tests.log
VTX_LIB=$PWD/videotransformer-pytorch_amd/libvtx_poolold.so timeout 300 python tools/micro/pool_bwd_bench.py > gpurun_out/r3i_pool_old.txt 2>&1
timeout 300 python tools/micro/pool_bwd_bench.py > gpurun_out/r3i_pool_new.txt 2>&1
paste gpurun_out/r3i_pool_old.txt gpurun_out/r3i_pool_new.txt
timeout 600 python tools/maskfeat_bench.py > gpurun_out/r3i_maskfeat.txt 2>&1; tail -2 gpurun_out/r3i_maskfeat.txt
