#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s19; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu_token_stationary" 2>&1 | tail -8 > $O/kernel_tests.txt
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu > $O/bench.txt
for v in pp0; do
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
done
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so timeout 300 python tools/geglu_trace.py 2>&1 | grep -v amdgpu > $O/trace.txt
timeout 600 bash tools/ab_libs.sh "default NS2VC_FUSE_GEGLU=0" "default NS2VC_FUSE_GEGLU=1" "pp0 NS2VC_FUSE_GEGLU=1" > $O/ab.txt 2>&1
tail -n 5 $O/kernel_tests.txt; cat $O/bench.txt $O/trace.txt $O/ab.txt
