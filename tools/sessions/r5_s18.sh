#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s18; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu > $O/bench.txt
for v in noslp prio noslp_prio gg1 gg2 gg4 gg8 gg14 gg15; do
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
done
for v in trace trace_noslp; do
  echo "## $v" >> $O/trace.txt
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/geglu_trace.py 2>&1 | grep -v amdgpu >> $O/trace.txt
done
cat $O/bench.txt $O/trace.txt
