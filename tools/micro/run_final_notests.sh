# end-of-round evidence without the test suite: profiles of the default command, stand-alone tables, other configurations
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/micro/profile_round.sh r2g > $O/r2g_profile_round.log 2>&1; tail -1 $O/r2g_bench.log | cut -c1-300
timeout 300 python tools/gemm_shapes.py 96 8 > $O/r2g_gemm_shapes.txt 2>&1
timeout 300 python tools/attn_bench.py 96 > $O/r2g_attn_bench.txt 2>&1
timeout 800 python tools/other_configs.py > $O/r2g_other_configs.txt 2>&1; cat $O/r2g_other_configs.txt | cut -c1-300
