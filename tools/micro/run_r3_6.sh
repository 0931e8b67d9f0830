set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu -k "ffn_skips or recompute" 2>&1 | tail -3 | cut -c1-300
bash tools/micro/profile_round3.sh r3g
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf -- python $R/tools/maskfeat_bench.py 32 3 > /tmp/mf.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_mf > $O/r3g_maskfeat_kernel_stats.csv
head -25 $O/r3g_maskfeat_kernel_stats.csv | cut -c1-200
