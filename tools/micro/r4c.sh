#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_nt or selftest" > gpurun_out/r4c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4c_tests.log
tail -3 gpurun_out/r4c_tests.log
( timeout 300 python tools/nt_ab.py 150528 2304 768 plain pp_epi=0,5,4
  timeout 300 python tools/nt_ab.py 150528 768 768 plain pp_epi=0,5,4
  timeout 300 python tools/nt_ab.py 150528 768 2304 plain pp_epi=0,5,4 ) 2>&1 | grep -v amdgpu > gpurun_out/r4c_ab.txt
cat gpurun_out/r4c_ab.txt
cd /tmp
for e in 0 5; do
  VTX_GEMM_PP_EPI=$e timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/prof_e$e -- python $GRAFT_REPO_ROOT/tools/nt_ab.py 150528 2304 768 plain gemm_nodma=0 1 10 > /tmp/log_e$e.txt 2>&1
  echo "pp_epi=$e" >> $GRAFT_REPO_ROOT/gpurun_out/r4c_pmc.txt
  python $GRAFT_REPO_ROOT/tools/pmc_dump.py /tmp/prof_e$e >> $GRAFT_REPO_ROOT/gpurun_out/r4c_pmc.txt 2>&1
done
cat $GRAFT_REPO_ROOT/gpurun_out/r4c_pmc.txt
