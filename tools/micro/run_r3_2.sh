# round 3, lease 3: pretrain tests; A/B on one box: dk/dv kernel variants, relaxed first-K-tile waits (pp_epi=4 switches them off)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_pretrain.py tests/test_gpu_kernels.py -q -m gpu -k "pretrain or attention or single_run" > $O/r3c_tests.log 2>&1; tail -5 $O/r3c_tests.log
for v in 0 1 2 3 4 0; do
  timeout 200 python tools/attn_bench.py 96 attn_dkv=$v 2>&1 | grep -E "options|spatial" | tr '\n' ' '; echo
done > $O/r3c_attn_dkv.txt 2>&1
cat $O/r3c_attn_dkv.txt
for e in 0 4 0 4; do
  VTX_GEMM_PP_EPI=$e timeout 300 python tools/gemm_shapes.py 96 8 2>&1 | grep -E "env|fc1 fwd|qkv_t fwd|qkv_s dgrad|fc1 dgrad|NT sum"
done > $O/r3c_relax_ab.txt 2>&1
cat $O/r3c_relax_ab.txt
