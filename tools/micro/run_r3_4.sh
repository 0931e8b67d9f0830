# round 3, lease 5: full GPU suite on the current build (packed staging, table row maps, FFN compaction, pretrain entry point),
# packed-vs-fp32 staging A/B, bench line
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
P=$R/videotransformer-pytorch_amd
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -q -m gpu > $O/r3e_gpu_tests.log 2>&1; tail -4 $O/r3e_gpu_tests.log | cut -c1-400; grep -E "^FAILED|^ERROR" $O/r3e_gpu_tests.log | cut -c1-300
for v in "" _f32st "" _f32st; do
  echo "== libvtx$v"
  VTX_LIB=$P/libvtx$v.so timeout 300 python tools/gemm_shapes.py 96 8 2>&1 | grep -E "^NT|NT sum"
done > $O/r3e_staging_ab.txt 2>&1
grep -E "==|NT sum" $O/r3e_staging_ab.txt
timeout 400 python bench.py --no-cpu-baseline > $O/r3e_bench.log 2>&1
tail -1 $O/r3e_bench.log | cut -c1-400
