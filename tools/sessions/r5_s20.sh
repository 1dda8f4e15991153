#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s20; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu > $O/bench.txt
for v in pp0 wt0 prio gg1 gg2 gg4 gg8 gg11 gg15; do
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
done
timeout 300 python tools/geglu_bench.py --rotate 1 2>&1 | grep -v amdgpu >> $O/bench.txt
cat $O/bench.txt
