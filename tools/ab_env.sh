#!/bin/bash
# same-box in-situ A/B of an environment switch of the library: bash tools/ab_env.sh VAR "v0 v1 ..." [rounds] > out
# prints, per run: VAR=value  ms/step  launches  graph==eager  isolated family ms
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
VAR=$1; VALS=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --skip-cpu --detail-json= --steps 20 --warmup 20 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$VAR=$v', round(d['ms_per_step'],4), d.get('launches_per_step'), d['graph_equals_eager'], {'gemm_family_ms_in_loop': d['roofline']['family_ms_in_loop']})
"
  done
done
