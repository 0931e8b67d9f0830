#!/bin/bash
# 8- and 32-clip steps: kernel time (rocprofv3 kernel trace) next to the wall time of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 8 32; do
  python $R/bench.py --batch $B --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown --no-other-configs > $O/r4k_b$B.json 2>$O/r4k_b$B.err
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_b$B -- python $R/bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-breakdown --no-other-configs > $O/r4k_prof_b$B.log 2>&1
  python $R/tools/rocpd_stats.py /tmp/prof_b$B > $O/r4k_kernel_stats_b$B.csv
  python - $B <<'PY'
import csv,sys,os,json
B=sys.argv[1]; O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
rows=list(csv.DictReader(open(O+f'r4k_kernel_stats_b{B}.csv')))
tot=sum(float(r['total_ns']) for r in rows); n=sum(int(r['calls']) for r in rows)
j=json.loads(open(O+f'r4k_b{B}.json').read().strip().splitlines()[-1])
print(f"B={B}: wall {j['ms_per_step']:.2f} ms/step {j['value']:.1f} clips/s; kernels {tot/7/1e6:.2f} ms/step in {n/7:.0f} launches")
for r in rows[:14]:
    print(f"  {float(r['total_ns'])/7/1e6:7.3f} ms {int(r['calls'])/7:6.1f}/step {float(r['avg_ns'])/1e3:8.1f} us  {r['kernel'][r['kernel'].find('vtx::')+5 if 'vtx::' in r['kernel'] else 0:][:80]}")
PY
done
