#!/bin/bash
# Round 6: same-box A/B of round 5's library against the current one on the step (no breakdown: vtx_hog_fwd changed its signature),
# and the kernel list of the 8-clip step.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp; P=$R/videotransformer-pytorch_amd
for i in 1 2 3; do for lib in libvtx_r5.so libvtx.so; do
  VTX_LIB=$P/$lib timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], 'nt', d['roofline']['avg_launch_us'])"
done; done > $O/r6i_lib_ab.txt; cat $O/r6i_lib_ab.txt
for i in 1 2; do for lib in libvtx_r5.so libvtx.so; do
  VTX_LIB=$P/$lib timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-breakdown 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('8 clips $lib', d['value'], d['ms_per_step'])"
done; done >> $O/r6i_lib_ab.txt; tail -4 $O/r6i_lib_ab.txt
cd /tmp; rm -rf /tmp/prof_b8
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_b8 -- python $R/bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-breakdown > /tmp/log_b8.txt 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_b8 > $O/r6i_kernel_stats_b8.csv; head -45 $O/r6i_kernel_stats_b8.csv | cut -c1-110,200-330
