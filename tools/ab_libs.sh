#!/bin/bash
# same-box A/B of library builds / plan switches inside the captured bench loop:  bash tools/ab_libs.sh "<lib|default> [ENV=VAL ...]" ...
# each argument is one configuration: a variant name under ns2vc_amd/lib/variants (or `default`) followed by environment assignments
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
cd "$(dirname "$0")/.."
CFGS=("$@")
for r in 1 2; do
  for cfg in "${CFGS[@]}"; do
    set -- $cfg; lib=$1; shift
    ( for kv in "$@"; do export "$kv"; done
      if [ "$lib" != default ]; then export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$lib/libns2vc_hip.so; fi
      timeout 300 python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l[:1]==chr(123)][-1])
print('$cfg', round(d['ms_per_step'],4), d.get('launches_per_step'), d['graph_equals_eager'], {'gemm_family_ms_in_loop': d['roofline']['family_ms_in_loop'], 'frac_isolated': d['roofline']['frac_isolated']})
" )
  done
done
