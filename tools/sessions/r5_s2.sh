#!/bin/bash
# round-5 GPU session 2: consumer fragment prefetch (default build) vs the compiler's order (variant pf0), tap-sharing on / off
cd "$(dirname "$0")/.."
O=gpurun_out/s2; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "tapshare or groupnorm_prologue or epilogue_groupnorm_stats or gemm_cases or shortcut or bench_shapes or heuristic or every_tile" 2>&1 | tail -8 > $O/kernel_tests.txt
timeout 400 python tools/gemm_sweep.py --ts --rotate 8 > $O/sweep_ts_rot8.txt 2>&1
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/pf0/libns2vc_hip.so timeout 400 python tools/gemm_sweep.py --ts --rotate 8 > $O/sweep_ts_rot8_pf0.txt 2>&1
timeout 1500 bash tools/ab_libs.sh "pf0 NS2VC_CONV_TS=0" "default NS2VC_CONV_TS=0" "default NS2VC_CONV_TS=1" "default NS2VC_CONV_TS=1 NS2VC_TS_NL=8" "default NS2VC_CONV_TS=1 NS2VC_TS_BN128_MIN=400" > $O/ab.txt 2>&1
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 --ops $O/ops_ts.txt > $O/bench_ts.json 2> $O/bench_ts.err
tail -n 3 $O/kernel_tests.txt; cat $O/ab.txt; cat $O/sweep_ts_rot8.txt
