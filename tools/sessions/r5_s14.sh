#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s14; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "tapshare or groupnorm_prologue or epilogue_groupnorm_stats or gemm_cases" 2>&1 | tail -12 > $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "tapshare or placement or block_by_block or golden or odd_shapes" 2>&1 | tail -12 > $O/engine_tests.txt
timeout 600 bash tools/ab_libs.sh "default NS2VC_CONV_WTILED=0" "default NS2VC_CONV_WTILED=1" > $O/ab.txt 2>&1
tail -n 4 $O/kernel_tests.txt $O/engine_tests.txt; cat $O/ab.txt
