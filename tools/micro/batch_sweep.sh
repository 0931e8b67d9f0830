#!/bin/bash
# Batch / frame-count sweep of the bench step with the RCCL exchange forced on the single rank (one box).  usage: bash tools/micro/batch_sweep.sh > out
for cfg in "8 8" "8 16" "8 32" "8 64" "8 96" "8 128" "16 8" "16 32" "16 48" "16 64"; do
  set -- $cfg
  VTX_FORCE_DP=1 python bench.py --frames $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-breakdown 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print(f\"frames $1 clips/GPU $2 (VTX_FORCE_DP=1): {d['value']:.1f} clips/s, {d['ms_per_step']:.2f} ms/step, NT GEMM roofline {r['frac']:.3f} ({r['avg_launch_us']:.0f} us/launch), nominal whole-step MFMA {d.get('mfma_frac_whole_step_nominal', float('nan')):.3f}\")"
done
