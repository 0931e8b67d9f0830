# end-of-round evidence: full GPU suite, profiles of the default command, stand-alone tables, other configurations
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/r2f_gpu_tests.log 2>&1; tail -3 $O/r2f_gpu_tests.log
bash tools/micro/profile_round.sh r2f > $O/r2f_profile_round.log 2>&1; tail -1 $O/r2f_bench.log | cut -c1-300
timeout 300 python tools/gemm_shapes.py 96 8 > $O/r2f_gemm_shapes.txt 2>&1
timeout 300 python tools/attn_bench.py 96 > $O/r2f_attn_bench.txt 2>&1
timeout 800 python tools/other_configs.py > $O/r2f_other_configs.txt 2>&1; cat $O/r2f_other_configs.txt | cut -c1-300
python - <<'PY' >> $O/r2f_other_configs.txt 2>&1
import sys, os
sys.argv = ['x']
sys.path.insert(0, os.path.join(os.environ['GRAFT_REPO_ROOT'], 'tools'))
import other_configs as oc
oc.timesformer_l96(batch=12)
PY
tail -2 $O/r2f_other_configs.txt | cut -c1-300
