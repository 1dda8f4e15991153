#!/bin/bash
# second session: LDS-path detectors (DETECT 2 / 3), full LDS wait (FIX 2), no packed fp32 (-fno-slp-vectorize)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=$PWD/ns2vc_amd/lib/variants
{
  for v in gnp_detect2 gnp_detect3 gnp_fix2 gnp_nopk gnp_base; do
    echo "== $v (SPEC tiles)"; GNP_REPS=10 NS2VC_LIB=$V/$v/libns2vc_hip.so timeout 300 python tools/gnp_probe.py 2>&1 | grep -v "x there\|columns differ\|amdgpu.ids" | cut -c1-260
  done
  for v in gnp_fix2 gnp_fz; do
    echo "== engine loop determinism, fuse_gn_gemm=1, $v"; NS2VC_FUSE_GN_GEMM=1 NS2VC_LIB=$V/$v/libns2vc_hip.so timeout 300 python tools/determinism_probe.py --steps 4 --more 6 --forwards 12 2>&1 | grep -v amdgpu.ids
  done
} 2>&1 | tee gpurun_out/gnp_rootcause2.txt
