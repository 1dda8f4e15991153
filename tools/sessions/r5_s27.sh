#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s27; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "token_stationary or plan_variants or golden" 2>&1 | tail -4 > $O/engine_tests.txt
for B in 1 4 8 16; do
  timeout 600 python bench.py --batch $B --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1]==chr(123)][-1])
print('batch', $B, 'ms/step', round(d['ms_per_step'],4), 'sample-steps/s', round($B*1000.0/d['ms_per_step'],1), 'launches', d.get('launches_per_step'), 'gemm-family frac', round(d['roofline']['frac'],4), 'isolated', round(d['roofline']['frac_isolated'],4), 'alone', d.get('gn_prologue_workgroups_alone'), 'graph==eager', d['loop_check']['graph_loop_equals_eager_loop'])
" >> $O/sweep.txt
done
tail -n 3 $O/engine_tests.txt; cat $O/sweep.txt
