#!/bin/bash
# Same-box A/B of two builds of the engine (box-to-box variance on the pool is +-3 %, run-to-run on one box +-0.3 %):
#   make -C ns2vc_amd/csrc OUT=../lib/variants/x -j4     # variant build from a patched source tree at the same depth
#   gpurun -- 'bash tools/ab_bench.sh ns2vc_amd/lib/variants/x/libns2vc_hip.so'
# alternates the default library (A) and the variant (B, via NS2VC_LIB) three times.
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
VAR=${1:?path of the variant libns2vc_hip.so}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = B ]; then export NS2VC_LIB=$PWD/$VAR; else unset NS2VC_LIB; fi
    python bench.py --skip-cpu --detail-json= --steps 20 --warmup 3 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), {'gemm_family_ms_in_loop': d['roofline']['family_ms_in_loop']})"
  done
done
