set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/r3f_gpu_tests.log 2>&1; tail -4 $O/r3f_gpu_tests.log | cut -c1-400; grep -E "^FAILED|^ERROR|Fatal" $O/r3f_gpu_tests.log | cut -c1-300
