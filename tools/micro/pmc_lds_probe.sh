cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $C | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $C -d /tmp/pmc_$tag -- python $R/tools/tn_compare.py > /tmp/log_$tag.txt 2>&1
  echo "== $C (tn_compare)" >> $R/gpurun_out/tn_pmc.txt
  python $R/tools/pmc_dump.py /tmp/pmc_$tag >> $R/gpurun_out/tn_pmc.txt 2>&1
  timeout 200 rocprofv3 --pmc $C -d /tmp/pmcn_$tag -- python $R/tools/gemm_hot.py > /tmp/logn_$tag.txt 2>&1
  echo "== $C (gemm_hot NT)" >> $R/gpurun_out/tn_pmc.txt
  python $R/tools/pmc_dump.py /tmp/pmcn_$tag >> $R/gpurun_out/tn_pmc.txt 2>&1
done
cat $R/gpurun_out/tn_pmc.txt
