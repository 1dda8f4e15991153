#!/bin/bash
# same-box comparison of the default build (A) and any number of variant libraries: bash tools/abn.sh name1 name2 ...
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
for i in 1 2; do
  for v in A "$@"; do
    if [ $v = A ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
    python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 3 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {k:round(v['ms_per_step_isolated'],3) for k,v in d['roofline']['families'].items() if k not in ('copy','other')})"
  done
done
