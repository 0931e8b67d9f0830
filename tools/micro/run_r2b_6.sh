set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > $O/r2b6_pytest.log 2>&1; tail -3 $O/r2b6_pytest.log
for o in "attn_w8=1" "attn_w8=0"; do
timeout 300 python tools/attn_bench.py 96 $o 2>&1 | grep "options\|attn"
done
