#!/bin/bash
# Round 5, first GPU call: where the red case stands (seed distribution, A/B of the wprod K-tile change), smoke, the WHOLE
# suite without -x (every margin into parity_report.txt), a baseline bench line of this binary.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R; export TMPDIR=/tmp
rm -f $O/parity_report.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -8 > $O/r5a_smoke.txt; cat $O/r5a_smoke.txt
timeout 300 python tests/diag_other_resolution.py hip 12 2>&1 | grep -v amdgpu.ids > $O/r5a_diag_default.txt
VTX_LIB=$R/videotransformer-pytorch_amd/libvtx_wp32.so timeout 300 python tests/diag_other_resolution.py hip 12 2>&1 | grep -v amdgpu.ids > $O/r5a_diag_wp32.txt
grep "seed   6" $O/r5a_diag_default.txt $O/r5a_diag_wp32.txt | cut -c1-220
rm -f $O/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q > $O/r5a_tests.log 2>&1
echo "tests rc=$?" >> $O/r5a_tests.log
tail -25 $O/r5a_tests.log | cut -c1-250
cp $O/parity_report.txt $O/r5a_parity_report.txt
grep -c . $O/r5a_parity_report.txt; grep FAIL $O/r5a_parity_report.txt | cut -c1-200
timeout 600 python bench.py > $O/r5a_bench.log 2>&1; tail -1 $O/r5a_bench.log | cut -c1-400
