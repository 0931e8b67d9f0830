#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $R/gpurun_out/r4c_tests.log 2>&1; grep -E "passed|failed" $R/gpurun_out/r4c_tests.log | tail -2
bash tools/micro/profile_round4b.sh r4c 2>&1 | tail -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
