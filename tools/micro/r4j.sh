#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_mf -- python $R/tools/maskfeat_bench.py 32 3 > $O/r4j_mf.log 2>&1
python $R/tools/rocpd_stats.py /tmp/prof_mf > $O/r4j_maskfeat_kernel_stats.csv
tail -1 $O/r4j_mf.log
python - <<'PY'
import csv,sys,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r4j_maskfeat_kernel_stats.csv')))
tot=sum(float(r['total_ns']) for r in rows)
print('total kernel ms per step', tot/5/1e6)
for r in rows[:40]:
    n=r['kernel'][:110]
    print(f"{float(r['total_ns'])/5/1e6:7.2f} ms  {int(r['calls'])//5:4d}/step  {float(r['avg_ns'])/1e3:8.1f} us  {n}")
PY
