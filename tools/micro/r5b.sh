#!/bin/bash
# Round 5, second GPU call: the one-wave-per-SIMD weight-gradient kernel (gemm_tn=w4): parity, A/B on the training shapes, step time.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -f $O/parity_report.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_tn or 65535 or streamed_kernels_vs_float64" > $O/r5b_tests.log 2>&1; echo "rc=$?" >> $O/r5b_tests.log
tail -15 $O/r5b_tests.log | cut -c1-300
timeout 300 python tools/tn_compare.py 150528 pp256 w4 2>&1 | grep -v amdgpu.ids > $O/r5b_tn_compare.txt; cat $O/r5b_tn_compare.txt
timeout 120 python tools/tn_compare.py 12544 pp256 w4 2>&1 | grep -v amdgpu.ids > $O/r5b_tn_compare_b8.txt; cat $O/r5b_tn_compare_b8.txt
for v in auto w4 auto w4; do
  VTX_GEMM_TN=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gemm_tn=$v', d['value'], d['ms_per_step'], d.get('gemm_tn_roofline'))" | cut -c1-400 | tee -a $O/r5b_bench_ab.txt
done
