# Round profiles of the default bench command (GPU box): kernel-trace stats + two separate PMC passes.
# usage: bash tools/micro/profile_round.sh <tag>     -> gpurun_out/<tag>_*
set -u
TAG=${1:-round}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $R/gpurun_out/${TAG}_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C -d /tmp/prof_$C -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/log_$C.txt 2>&1
  python $R/tools/pmc_dump.py /tmp/prof_$C > $R/gpurun_out/${TAG}_pmc_$C.txt
done
cd $R
timeout 600 python bench.py > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
cat gpurun_out/${TAG}_pmc_FETCH_SIZE.txt gpurun_out/${TAG}_pmc_WRITE_SIZE.txt
