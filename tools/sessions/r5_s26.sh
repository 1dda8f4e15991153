#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s26; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu_token_stationary" 2>&1 | tail -4 > $O/kernel_tests.txt
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu > $O/bench.txt
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/rot0/libns2vc_hip.so timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
timeout 600 bash tools/ab_libs.sh "rot0" "default" > $O/ab.txt 2>&1
tail -n 3 $O/kernel_tests.txt; cat $O/bench.txt $O/ab.txt
