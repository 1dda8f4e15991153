#!/bin/bash
# round-5 GPU session 1: the tap-sharing conv kernel -- tests, isolated sweep, in-situ A/B, per-launch table
cd "$(dirname "$0")/.."
O=gpurun_out/s1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "tapshare or groupnorm_prologue or epilogue_groupnorm_stats or gemm_cases or shortcut or bench_shapes or heuristic" 2>&1 | tail -25 > $O/kernel_tests.txt
cp gpurun_out/test_diag.txt $O/kernel_diag.txt 2>/dev/null
timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -25 > $O/engine_tests.txt
cp gpurun_out/test_diag.txt $O/engine_diag.txt 2>/dev/null
timeout 600 python tools/gemm_sweep.py --ts --rotate 8 > $O/sweep_ts_rot8.txt 2>&1
timeout 1500 bash tools/ab_libs.sh "default NS2VC_CONV_TS=0" "default NS2VC_CONV_TS=1" "default NS2VC_CONV_TS=1 NS2VC_TS_NL=8" "default NS2VC_CONV_TS=1 NS2VC_TS_BN128_MIN=100" "default NS2VC_CONV_TS=1 NS2VC_TS_BN128_MIN=400" > $O/ab_ts.txt 2>&1
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 --ops $O/ops_ts.txt > $O/bench_ts.json 2> $O/bench_ts.err
tail -5 $O/kernel_tests.txt $O/engine_tests.txt; cat $O/ab_ts.txt
