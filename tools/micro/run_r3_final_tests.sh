#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r3k_gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3k_gpu_tests.log
tail -5 gpurun_out/r3k_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
