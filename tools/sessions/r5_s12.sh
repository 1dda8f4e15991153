#!/bin/bash
# r5 session 12: ablations of conv3ts_kernel's step (variant builds, wrong results, timing only): what bounds the K loop?
cd "$(dirname "$0")/.."
O=gpurun_out/s12; mkdir -p $O
export TMPDIR=/tmp
for v in default abl1 abl2 abl4; do
  echo "## $v (1: consumers idle  2: no DMA inside the loop  4: fragment reads without MFMAs)" >> $O/ablate.txt
  if [ $v = default ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
  timeout 300 python tools/gemm_sweep.py --ts --rotate 8 2>&1 | grep -v amdgpu >> $O/ablate.txt
done
cat $O/ablate.txt
