#!/bin/bash
# Round-6 evidence set of ONE binary on ONE box.  usage: bash tools/micro/profile_round6.sh <tag>
# Order: profiles and bench lines first, the WHOLE GPU suite (exactly the driver's command, -x included) LAST on the same tree --
# its log and the parity report with every check() margin go to gpurun_out/<tag>_* and from there into profiles/ (VERDICT r4 item 1).
set -u
TAG=${1:-r6}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $O/${TAG}_rocprofv3_kernel_stats_b96.csv
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_exact -- python $R/bench.py --stream fp32 --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_exact_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt_exact > $O/${TAG}_rocprofv3_kernel_stats_exact_b96.csv
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt_exact_grad -- python $R/bench.py --stream fp32+grad --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_exact_grad_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt_exact_grad > $O/${TAG}_rocprofv3_kernel_stats_exact_grad_b96.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C -d /tmp/prof_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_$C.txt 2>&1
  python $R/tools/pmc_dump.py /tmp/prof_$C > $O/${TAG}_pmc_${C}_b96.txt
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_mfma.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_mfma > $O/${TAG}_pmc_MFMA_BUSY_b96.txt
cd $R
timeout 900 python bench.py > $O/${TAG}_bench_default_b96.log 2>&1; tail -1 $O/${TAG}_bench_default_b96.log | cut -c1-300
VTX_FORCE_DP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-other-configs > $O/${TAG}_bench_force_dp_b96.log 2>&1; tail -1 $O/${TAG}_bench_force_dp_b96.log | cut -c1-200
timeout 300 python bench.py --stream fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_exact_stream_b96.log 2>&1; tail -1 $O/${TAG}_bench_exact_stream_b96.log | cut -c1-200
timeout 300 python bench.py --stream fp32+grad --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_exact_grad_stream_b96.log 2>&1; tail -1 $O/${TAG}_bench_exact_grad_stream_b96.log | cut -c1-200
timeout 300 python tools/hog_bench.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_hog.txt; cat $O/${TAG}_hog.txt
timeout 600 python tools/other_configs.py vivit tsf16 tsfl96_stored tsfl96_12 > $O/${TAG}_other_configs.txt 2>&1; cut -c1-250 $O/${TAG}_other_configs.txt | grep -v amdgpu.ids
timeout 300 python tools/maskfeat_bench.py 32 3 > $O/${TAG}_maskfeat.txt 2>&1; tail -1 $O/${TAG}_maskfeat.txt | cut -c1-250
# LAST: the driver's command on this tree
rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_suite.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_gpu_suite.log
tail -5 $O/${TAG}_gpu_suite.log | cut -c1-250
cp $O/parity_report.txt $O/${TAG}_parity_report.txt
echo "parity lines: $(grep -c . $O/${TAG}_parity_report.txt), FAIL lines: $(grep -c '^FAIL' $O/${TAG}_parity_report.txt)"
( cd $R && git rev-parse HEAD 2>/dev/null; sha256sum videotransformer-pytorch_amd/libvtx.so bench.py | cut -c1-80 ) > $O/${TAG}_tree.txt 2>&1; cat $O/${TAG}_tree.txt
