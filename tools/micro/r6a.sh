#!/bin/bash
# Round 6, first GPU call: this box's baseline of round 5's binary + the three measurements VERDICT r5 asked for (items 4, 5, 8)
# + the INTEGRATION.md stub test.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "integration_stub or selftest" > $O/r6a_tests.log 2>&1; echo "rc=$?" >> $O/r6a_tests.log; tail -3 $O/r6a_tests.log
timeout 600 python bench.py > $O/r6a_bench.log 2>&1; tail -1 $O/r6a_bench.log > $O/r6a_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6a_bench.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d.get('gemm_tn_roofline'))
for k in ('nt', 'tn'):
    for r in d['gemm_shapes'][k]: print(k, r['shape'], r['launches_per_step'], r['avg_us'], r['frac'])
for r in d.get('roofline_hbm', []): print('hbm', r.get('kernel', '')[:50], r.get('frac'), r.get('avg_launch_us'), r.get('ms_per_step'))
for r in d.get('other_configs', []): print('other', r.get('workload', '')[:70], r.get('clips_per_s'), r.get('ms_per_step'), r.get('skipped'), r.get('error'))
PY
timeout 600 python tools/blaslt_ref.py 96 2>&1 | grep -v amdgpu.ids > $O/r6a_blaslt_ref.txt; cat $O/r6a_blaslt_ref.txt
timeout 600 python tools/cu_contention.py 2>&1 | grep -v amdgpu.ids > $O/r6a_cu_contention.txt; cat $O/r6a_cu_contention.txt
timeout 300 python tools/gemm_shapes.py 96 2>&1 | grep -v amdgpu.ids > $O/r6a_gemm_shapes.txt; cat $O/r6a_gemm_shapes.txt
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_clk -- python $R/tools/blaslt_ref.py 96 --rounds 2 --launches 10 > /tmp/log_clk.txt 2>&1
python $R/tools/pmc_dump.py /tmp/prof_clk gemm,Cijk,Custom --clock --schema > $O/r6a_blaslt_clock.txt 2>&1; python $R/tools/rocpd_stats.py /tmp/prof_clk | head -30 >> $O/r6a_blaslt_clock.txt; cat $O/r6a_blaslt_clock.txt | cut -c1-230
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d /tmp/prof_hog1 -- python $R/tools/hog_bench.py > /tmp/log_hog1.txt 2>&1; tail -2 /tmp/log_hog1.txt
python $R/tools/pmc_dump.py /tmp/prof_hog1 hog > $O/r6a_hog_pmc.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof_hog2 -- python $R/tools/hog_bench.py > /tmp/log_hog2.txt 2>&1; tail -2 /tmp/log_hog2.txt
python $R/tools/pmc_dump.py /tmp/prof_hog2 hog >> $O/r6a_hog_pmc.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/prof_hog3 -- python $R/tools/hog_bench.py > /tmp/log_hog3.txt 2>&1; tail -2 /tmp/log_hog3.txt
python $R/tools/pmc_dump.py /tmp/prof_hog3 hog >> $O/r6a_hog_pmc.txt 2>&1
cat $O/r6a_hog_pmc.txt
