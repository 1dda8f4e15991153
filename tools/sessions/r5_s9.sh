#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s9; mkdir -p $O
export TMPDIR=/tmp
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so timeout 200 python tools/ts_trace.py --gnp > $O/ts_trace_gnp.txt 2>&1
timeout 600 bash tools/ab_libs.sh "default" "default NS2VC_TS_BN128_MIN=200" "default NS2VC_TS_NL=4" > $O/ab.txt 2>&1
grep -v amdgpu $O/ts_trace_gnp.txt; cat $O/ab.txt
