#!/bin/bash
# round-5 GPU session 4: role-specialised tap-sharing loops + relative placement check; phase trace
cd "$(dirname "$0")/.."
O=gpurun_out/s5; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "tapshare or groupnorm_prologue or epilogue_groupnorm_stats or gemm_cases or shortcut or bench_shapes or heuristic" 2>&1 | tail -12 > $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "tapshare or placement or env_switches or block_by_block or ragged or odd_shapes or bit_identical" 2>&1 | tail -12 > $O/engine_tests.txt
cp gpurun_out/test_diag.txt $O/engine_diag.txt 2>/dev/null
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so timeout 300 python tools/ts_trace.py > $O/ts_trace.txt 2>&1
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/trace/libns2vc_hip.so timeout 300 python tools/ts_trace.py --gnp > $O/ts_trace_gnp.txt 2>&1
timeout 400 python tools/gemm_sweep.py --ts --rotate 8 > $O/sweep_ts_rot8.txt 2>&1
timeout 300 python tools/r5_s3.py > $O/diag.txt 2>&1
timeout 900 bash tools/ab_libs.sh "default NS2VC_CONV_TS=0" "default NS2VC_CONV_TS=1" "default NS2VC_CONV_TS=1 NS2VC_TS_NL=8" > $O/ab.txt 2>&1
tail -n 4 $O/kernel_tests.txt $O/engine_tests.txt; cat $O/ab.txt; cat $O/ts_trace.txt $O/ts_trace_gnp.txt; grep -v amdgpu $O/diag.txt | tail -12; cat $O/sweep_ts_rot8.txt
