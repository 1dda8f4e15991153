#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s13; mkdir -p $O
export TMPDIR=/tmp
for v in default wtiled; do
  echo "## $v" >> $O/wtiled.txt
  if [ $v = default ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
  timeout 300 python tools/gemm_sweep.py --ts --rotate 8 2>&1 | grep -v amdgpu >> $O/wtiled.txt
  timeout 300 python tools/gemm_sweep.py --ts --rotate 40 2>&1 | grep -v amdgpu >> $O/wtiled.txt
done
unset NS2VC_LIB
timeout 600 bash tools/ab_libs.sh "default" "wtiled" >> $O/wtiled.txt 2>&1
cat $O/wtiled.txt
