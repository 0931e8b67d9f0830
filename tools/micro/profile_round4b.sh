# Round-4 evidence, second set (after the streamed attention backward).  usage: bash tools/micro/profile_round4b.sh <tag>
set -u
TAG=${1:-r4b}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $O/${TAG}_rocprofv3_kernel_stats_b96.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C -d /tmp/prof_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_$C.txt 2>&1
  python $R/tools/pmc_dump.py /tmp/prof_$C > $O/${TAG}_pmc_${C}_b96.txt
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_mfma.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_mfma > $O/${TAG}_pmc_MFMA_BUSY_b96.txt
cd $R
timeout 600 python bench.py > $O/${TAG}_bench_default_b96.log 2>&1; tail -1 $O/${TAG}_bench_default_b96.log | cut -c1-300
VTX_FORCE_DP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-other-configs > $O/${TAG}_bench_force_dp_b96.log 2>&1
# attention backward: the three forms back to back, twice
( for i in 1 2; do for f in 0 1 2; do echo -n "attn_fused=$f: "; timeout 120 python tools/attn_bench.py 96 attn_fused=$f 2>&1 | grep "bwd spatial"; done; done
  for f in 1 2; do echo -n "8 clips attn_fused=$f: "; timeout 120 python tools/attn_bench.py 8 attn_fused=$f 2>&1 | grep "bwd spatial"; done ) > $O/${TAG}_attn_bwd_stream.txt 2>&1
python videotransformer-pytorch_amd/csrc/build.py --variant trace VTX_STREAM_TRACE=1 > /dev/null 2>&1
VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_trace.so timeout 120 python tools/attn_timeline.py 96 2>&1 | grep -v -i "warning\|amdgpu.ids" >> $O/${TAG}_attn_bwd_stream.txt
cat $O/${TAG}_attn_bwd_stream.txt | head -12
timeout 900 python tools/other_configs.py vivit tsf16 tsfl96_stored tsfl96_12 > $O/${TAG}_other_configs.txt 2>&1; cut -c1-250 $O/${TAG}_other_configs.txt
timeout 300 python tools/maskfeat_bench.py 32 3 > $O/${TAG}_maskfeat.txt 2>&1; tail -1 $O/${TAG}_maskfeat.txt
# the bf16 margin of the headline model (per-commit tracking asked for by VERDICT r3): worst gradient of the T = 8 training golden
rm -f $O/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -k "t8_train_vs_golden" 2>&1 | tail -1
grep "parameter gradients" $O/parity_report.txt > $O/${TAG}_bf16_worst_gradient.txt; cat $O/${TAG}_bf16_worst_gradient.txt | cut -c1-250
