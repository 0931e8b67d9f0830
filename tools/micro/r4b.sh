#!/bin/bash
# round 4, call B: rolling front half of the plain epilogue -- parity, same-box A/B (pp_epi 0 = rolling, 5 = lean behind the loop, 4 = general passes)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or selftest" > gpurun_out/r4b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4b_tests.log
tail -4 gpurun_out/r4b_tests.log
for rep in 1 2; do
  for e in 0 5; do
    VTX_GEMM_PP_EPI=$e timeout 300 python tools/gemm_shapes.py 96 8 > gpurun_out/r4b_shapes_epi${e}_$rep.txt 2>&1
  done
done
grep "NT sum" gpurun_out/r4b_shapes_*.txt
for e in 0 5; do
  for k in plain; do
    timeout 120 python tools/pp_timeline.py 150528 2304 768 $k pp_epi=$e >> gpurun_out/r4b_timeline.txt 2>&1
  done
done
grep -v amdgpu gpurun_out/r4b_timeline.txt
