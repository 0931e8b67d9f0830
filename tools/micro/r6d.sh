#!/bin/bash
# wprod kernel times from the kernel trace (the Python loop of tools/wprod_bench.py is host-bound at ~23 us per call)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in libvtx.so libvtx_wpp0.so libvtx_wpp56.so; do
  rm -rf /tmp/prof_w
  VTX_LIB=$R/videotransformer-pytorch_amd/$L timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_w -- python $R/tools/wprod_bench.py > /tmp/log_w.txt 2>&1
  echo "== $L"; python $R/tools/rocpd_stats.py /tmp/prof_w | grep -i wprod 
done > $O/r6d_wprod_trace.txt 2>&1; cat $O/r6d_wprod_trace.txt
