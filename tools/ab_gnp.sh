#!/bin/bash
# GroupNorm prologue A/B: default lib and variant libraries, each with the fusion off / on
for r in 1 2; do
  for lib in default "$@"; do
    for v in 0 1; do
      if [ $lib = default ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$lib/libns2vc_hip.so; fi
      NS2VC_FUSE_GN_GEMM=$v python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 20 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$lib fuse=$v', round(d['ms_per_step'],4), d.get('launches_per_step'), d['loop_check'], {k: v2['ms_per_step_isolated'] for k,v2 in d['roofline']['families'].items() if k in ('implicit_gemm','norm_stats')})
"
    done
  done
done
