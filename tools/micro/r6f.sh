#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_wptrace.so python tools/micro/wprod_timeline.py 2>&1 | grep -v amdgpu > $O/r6f_wprod_timeline.txt; cat $O/r6f_wprod_timeline.txt
cd /tmp
for L in libvtx.so libvtx_wpn1.so; do
  rm -rf /tmp/prof_w
  VTX_LIB=$R/videotransformer-pytorch_amd/$L timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_w -- python $R/tools/wprod_bench.py > /tmp/log_w.txt 2>&1
  echo "== $L"; python $R/tools/rocpd_stats.py /tmp/prof_w | grep -i wprod
done > $O/r6f_wprod_trace.txt 2>&1; cat $O/r6f_wprod_trace.txt
cd $R; timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "wprod or merged" 2>&1 | tail -2
