#!/bin/bash
# r5 session 16: geglu.hip with the software-pipelined GEGLU: correctness, isolated time, ablations (diagnostic builds: timing only)
cd "$(dirname "$0")/.."
O=gpurun_out/s16; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu_token_stationary" 2>&1 | tail -8 > $O/kernel_tests.txt
timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu > $O/bench.txt
for v in ggs0 gg1 gg2 gg4 gg8 gg3 gg6 gg14; do
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/geglu_bench.py 2>&1 | grep -v amdgpu >> $O/bench.txt
done
tail -n 4 $O/kernel_tests.txt; cat $O/bench.txt
