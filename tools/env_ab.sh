#!/bin/bash
# same-box A/B of HIP runtime switches that act on the kernel-to-kernel path of a captured graph (210 dependent launches per step).
# Every run is under its own `timeout`: ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the process on this image (it cost a 15-minute GPU call)
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
run() {
  env "$@" timeout 120 python bench.py --skip-cpu --detail-json= --steps 20 --warmup 3 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],4), round(min(d['jobs_ms'])/d['steps'],4), d['graph_equals_eager'])"
}
for i in 1 2; do
  run X=0
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run AMD_OPT_FLUSH=0
  run AMD_OPT_FLUSH=1
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run ROC_USE_FGS_KERNARG=0
  run ROC_USE_FGS_KERNARG=1
done
