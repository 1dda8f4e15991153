#!/bin/bash
# r5 session 23: rowchain stage-2 results staged through LDS (1-KB runs), attention results in 16-byte pieces: tests, same-box A/B against the previous build
cd "$(dirname "$0")/.."
O=gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rowchain or attention" 2>&1 | tail -8 > $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "block_by_block or golden or odd_shapes or sliced or token_stationary" 2>&1 | tail -8 > $O/engine_tests.txt
timeout 900 bash tools/ab_libs.sh "prev" "rc_only" "default" > $O/ab.txt 2>&1
tail -n 4 $O/kernel_tests.txt $O/engine_tests.txt; cat $O/ab.txt
