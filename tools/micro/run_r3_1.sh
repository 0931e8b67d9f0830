# round 3, first lease: full GPU suite + kernel-trace stats of the bench (are the rocBLAS / ATen-CE launches gone?) + bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu  > $O/r3b_gpu_tests.log 2>&1; tail -15 $O/r3b_gpu_tests.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/r3b_bench_under_rocprof.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_kt > $O/r3b_kernel_stats.csv
cd $R
grep -v "vtx::" $O/r3b_kernel_stats.csv | cut -c1-150
timeout 400 python bench.py --no-cpu-baseline > $O/r3b_bench.log 2>&1
tail -1 $O/r3b_bench.log | cut -c1-600
