# Round-4 evidence of the default bench command and the other configurations (GPU box).  usage: bash tools/micro/profile_round4.sh <tag>
#   -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/round3_*)
set -u
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace of the default command (3 timed + 2 warm-up + 2 instrumented steps)
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $O/${TAG}_rocprofv3_kernel_stats_b96.csv
# 2. counters: separate passes, no tracing
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C -d /tmp/prof_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_$C.txt 2>&1
  python $R/tools/pmc_dump.py /tmp/prof_$C > $O/${TAG}_pmc_${C}_b96.txt
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_mfma -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-breakdown --no-other-configs > /tmp/log_mfma.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_mfma > $O/${TAG}_pmc_MFMA_BUSY_b96.txt
cd $R
# 3. the bench line (with the CPU baseline) and the forced-RCCL variant
timeout 600 python bench.py > $O/${TAG}_bench_default_b96.log 2>&1; tail -1 $O/${TAG}_bench_default_b96.log | cut -c1-300
VTX_FORCE_DP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown > $O/${TAG}_bench_force_dp_b96.log 2>&1
# 4. batch sweep, 8 and 16 frames, each with the RCCL exchange forced on one rank
( for F in 8 16; do for B in 8 32 96; do
    if [ $F = 16 ] && [ $B = 96 ]; then B=48; fi
    VTX_FORCE_DP=1 timeout 300 python bench.py --frames $F --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-breakdown --no-other-configs 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(f\"frames $F clips/GPU $B (VTX_FORCE_DP=1): {d['value']:.1f} clips/s, {d['ms_per_step']:.2f} ms/step, NT GEMM roofline {d['roofline']['frac']:.3f} ({d['roofline']['avg_launch_us']:.0f} us/launch), nominal whole-step MFMA {d['mfma_frac_whole_step_nominal']:.3f}\")"
  done; done ) > $O/${TAG}_batch_sweep.txt 2>&1
cat $O/${TAG}_batch_sweep.txt
# 5. other BASELINE configurations
timeout 900 python tools/other_configs.py hog vivit tsf16 tsfl96_stored tsfl96_12 > $O/${TAG}_other_configs.txt 2>&1; cut -c1-250 $O/${TAG}_other_configs.txt
timeout 300 python tools/maskfeat_bench.py 32 3 > $O/${TAG}_maskfeat.txt 2>&1; tail -1 $O/${TAG}_maskfeat.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf -- python $R/tools/maskfeat_bench.py 32 3 > /tmp/mf.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_mf > $O/${TAG}_maskfeat_kernel_stats.csv
cd $R
