# round 3, lease 4: build-time experiments of the persistent NT GEMM, A/B on one box through VTX_LIB
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
P=$R/videotransformer-pytorch_amd
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_pretrain.py -q -m gpu > $O/r3d_pretrain.log 2>&1; tail -3 $O/r3d_pretrain.log | cut -c1-300
for v in iss1 iss2 iss3 bf16st; do
  VTX_LIB=$P/libvtx_$v.so timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm_nt" 2>&1 | tail -1 | sed "s/^/$v: /"
done > $O/r3d_variant_tests.txt 2>&1
cat $O/r3d_variant_tests.txt
for v in "" _iss1 _iss2 _iss3 _bf16st "" _iss3 _bf16st; do
  echo "== libvtx$v"
  VTX_LIB=$P/libvtx$v.so timeout 300 python tools/gemm_shapes.py 96 8 2>&1 | grep -E "^NT|NT sum"
done > $O/r3d_variants.txt 2>&1
grep -E "==|NT sum" $O/r3d_variants.txt
