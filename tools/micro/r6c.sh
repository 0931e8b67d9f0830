#!/bin/bash
# Round 6, third GPU call: look-ahead depth of the wprod kernel (variants), HOG table-size check, CPU-side things that need the GPU.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
for rep in 1 2; do
for L in libvtx.so libvtx_wpp0.so libvtx_wpp56.so; do
  VTX_LIB=$R/videotransformer-pytorch_amd/$L timeout 300 python tools/wprod_bench.py 2>&1 | grep -v amdgpu.ids
done
done > $O/r6c_wprod.txt; cat $O/r6c_wprod.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "hog or selftest" > $O/r6c_tests.log 2>&1; echo "rc=$?" >> $O/r6c_tests.log; tail -3 $O/r6c_tests.log
