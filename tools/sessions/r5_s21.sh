#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s21; mkdir -p $O
export TMPDIR=/tmp
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so timeout 300 python tools/geglu_trace.py 2>&1 | grep -v amdgpu > $O/trace.txt
cat $O/trace.txt
