#!/bin/bash
# r4 root-cause session for the GroupNorm-prologue non-determinism (branch gnp-rootcause): standalone repro + instrumented variants.
#   variants: make -C ns2vc_amd/csrc OUT=../lib/variants/<name> DEFS=...   (gnp_base, gnp_detect, gnp_fix1, gnp_fz)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/ns2vc_amd/lib/variants
{
  echo "== standalone repro (tools/vmem_dma_race.hip)"; timeout 120 tools/bin/vmem_dma_race
  for v in gnp_base gnp_detect gnp_fix1 gnp_fz; do
    echo "== $v (SPEC tiles)"; NS2VC_LIB=$V/$v/libns2vc_hip.so timeout 300 python tools/gnp_probe.py
  done
  echo "== gnp_base, plain tiles (NS2VC_GEMM_SPEC=0)"; NS2VC_GEMM_SPEC=0 NS2VC_LIB=$V/gnp_base/libns2vc_hip.so timeout 300 python tools/gnp_probe.py
  for v in gnp_base gnp_fix1; do
    echo "== engine loop determinism, fuse_gn_gemm=1, $v"; NS2VC_FUSE_GN_GEMM=1 NS2VC_LIB=$V/$v/libns2vc_hip.so timeout 300 python tools/determinism_probe.py --steps 4 --more 4
  done
  echo "== box baseline (gnp_base, fusion off)"; NS2VC_LIB=$V/gnp_base/libns2vc_hip.so timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d.get('launches_per_step'))"
} 2>&1 | tee gpurun_out/gnp_rootcause.txt
