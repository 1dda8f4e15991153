#!/bin/bash
# third session: compiler's packed products vs scalar products of the same registers (DETECT 4); in-situ cost of -fno-slp-vectorize and of the fix
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/ns2vc_amd/lib/variants
{
  echo "== gnp_detect4 (SPEC tiles)"; GNP_REPS=10 NS2VC_LIB=$V/gnp_detect4/libns2vc_hip.so timeout 300 python tools/gnp_probe.py 2>&1 | grep -v "x there\|columns differ\|amdgpu.ids" | cut -c1-330
  for r in 1 2; do
    for cfg in "gnp_base 0" "gnp_nopk 0" "gnp_nopk 1" "gnp_fix2 1" "gnp_fz 1"; do
      set -- $cfg
      NS2VC_FUSE_GN_GEMM=$2 NS2VC_LIB=$V/$1/libns2vc_hip.so timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$1 fuse=$2', round(d['ms_per_step'],4), d.get('launches_per_step'), d['loop_check'], {k: round(v2['ms_per_step_isolated'],4) for k,v2 in d['roofline']['families'].items() if k in ('implicit_gemm','norm_stats','attention')})
"
    done
  done
} 2>&1 | tee gpurun_out/gnp_rootcause3.txt
