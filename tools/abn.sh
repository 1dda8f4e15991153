#!/bin/bash
# same-box comparison of the default build (A) and any number of variant libraries: bash tools/abn.sh name1 name2 ...
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
for i in 1 2; do
  for v in A "$@"; do
    if [ $v = A ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
    python bench.py --skip-cpu --detail-json= --steps 20 --warmup 3 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {'gemm_family_ms_in_loop': d['roofline']['family_ms_in_loop']})"
  done
done
