set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > $O/r2b5_pytest.log 2>&1; tail -3 $O/r2b5_pytest.log
VTX_ATTN_HW_FWD=0 VTX_ATTN_HW_BWD=6 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention" > $O/r2b5_pytest_hw.log 2>&1; tail -3 $O/r2b5_pytest_hw.log
for o in "attn_hw_bwd=0" "attn_hw_bwd=3" "attn_hw_bwd=4" "attn_hw_bwd=6" "attn_hw_bwd=12" "attn_hw_fwd=6" "attn_hw_fwd=4"; do
timeout 300 python tools/attn_bench.py 96 $o 2>&1 | grep "options\|temporal"
done
