set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_mvit.py tests/test_gpu_models.py -q -m gpu -k "cross_attention or mvit or maskfeat or ffn_skips" > $O/r3h_tests.log 2>&1; tail -3 $O/r3h_tests.log | cut -c1-300; grep -E "^FAILED|^ERROR|Fatal" $O/r3h_tests.log | cut -c1-250
timeout 300 python tools/maskfeat_bench.py 32 3 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf -- python $R/tools/maskfeat_bench.py 32 3 > /tmp/mf.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_mf > $O/r3h_maskfeat_kernel_stats.csv
head -14 $O/r3h_maskfeat_kernel_stats.csv | cut -c1-60,200-260
