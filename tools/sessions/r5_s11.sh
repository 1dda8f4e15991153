#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s11; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "tapshare or groupnorm_prologue or epilogue_groupnorm_stats" 2>&1 | tail -8 > $O/kernel_tests.txt
timeout 400 python tools/gemm_sweep.py --ts --rotate 8 > $O/sweep_ts_rot8.txt 2>&1
timeout 900 bash tools/ab_libs.sh "default" "default NS2VC_TS_KS=1" "default NS2VC_GN_COOP_MIN=3" "default NS2VC_TS_KS=1 NS2VC_GN_COOP_MIN=3" "default NS2VC_GN_COOP_MIN=4" > $O/ab.txt 2>&1
tail -n 3 $O/kernel_tests.txt; cat $O/ab.txt; grep -v amdgpu $O/sweep_ts_rot8.txt
