#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/micro/profile_round4b.sh r4b 2>&1 | tail -40
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
