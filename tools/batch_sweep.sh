#!/bin/bash
# the bench job at batch 1 .. 128 on one box:  bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt   (r4 / r5: profiles/rNN_batch_sweep.txt)
cd "$(dirname "$0")/.."
for B in 1 4 8 16 32 64 128; do
  timeout 600 python bench.py --batch $B --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1]==chr(123)][-1])
print('batch', $B, 'ms/step', round(d['ms_per_step'],4), 'sample-steps/s', round($B*1000.0/d['ms_per_step'],1), 'launches', d.get('launches_per_step'), 'gemm-family frac', round(d['roofline']['frac'],4), 'isolated', round(d['roofline']['frac_isolated'],4), 'graph==eager', d['graph_equals_eager'])
"
done
