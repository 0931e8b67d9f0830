#!/bin/bash
# round 4, call A: lean passes -- parity of the NT GEMM tests, same-box A/B of the layer's GEMMs (lean vs general passes), bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or selftest" > gpurun_out/r4a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4a_tests.log
tail -5 gpurun_out/r4a_tests.log
for rep in 1 2; do
  timeout 300 python tools/gemm_shapes.py 96 8 > gpurun_out/r4a_shapes_lean_$rep.txt 2>&1
  VTX_GEMM_PP_EPI=4 timeout 300 python tools/gemm_shapes.py 96 8 > gpurun_out/r4a_shapes_old_$rep.txt 2>&1
done
grep "NT sum" gpurun_out/r4a_shapes_*.txt
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r4a_bench.log 2>&1
tail -c 1500 gpurun_out/r4a_bench.log
for k in plain residual gelu2 mul; do
  timeout 120 python tools/pp_timeline.py 150528 768 768 $k >> gpurun_out/r4a_timeline.txt 2>&1
done
