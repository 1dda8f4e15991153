#!/bin/bash
# where does the token-stationary GEGLU kernel start to pay?  batch 16 (3760 rows at level 2) and batch 24 (5640), GEMM vs kernel, same box
cd "$(dirname "$0")/.."
O=gpurun_out/s28; mkdir -p $O
export TMPDIR=/tmp NS2VC_DEBUG_ENV=1
for B in 16 24; do for G in 0 1; do
  NS2VC_FUSE_GEGLU=$G timeout 200 python bench.py --batch $B --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1]==chr(123)][-1])
print('batch', $B, 'fuse_geglu', $G, 'ms/step', round(d['ms_per_step'],4))
" >> $O/cross.txt
done; done
cat $O/cross.txt
