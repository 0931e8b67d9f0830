#!/bin/bash
# counters of the wprod kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_w1 /tmp/prof_w2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/prof_w1 -- python $R/tools/wprod_bench.py > /tmp/log_w1.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_w1 wprod > $O/r6e_wprod_pmc.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/prof_w2 -- python $R/tools/wprod_bench.py > /tmp/log_w2.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_w2 wprod >> $O/r6e_wprod_pmc.txt 2>&1
tail -3 /tmp/log_w2.txt
rm -rf /tmp/prof_w3
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d /tmp/prof_w3 -- python $R/tools/wprod_bench.py > /tmp/log_w3.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_w3 wprod >> $O/r6e_wprod_pmc.txt 2>&1
tail -3 /tmp/log_w3.txt
cat $O/r6e_wprod_pmc.txt
