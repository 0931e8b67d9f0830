#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
true
rm -f $O/parity_report.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $O/r6h_gpu_suite.log 2>&1; echo "pytest rc=$?" >> $O/r6h_gpu_suite.log
tail -5 $O/r6h_gpu_suite.log | cut -c1-250
cp $O/parity_report.txt $O/r6h_parity_report.txt
echo "parity lines: $(grep -c . $O/r6h_parity_report.txt), FAIL lines: $(grep -c '^FAIL' $O/r6h_parity_report.txt)"
