#!/bin/bash
# round-5 GPU session 6: full GPU suite on the tap-sharing build, loader-wave / prefetch A/B, per-launch table, CU partition measurement
cd "$(dirname "$0")/.."
O=gpurun_out/s6; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.txt
cp gpurun_out/test_diag.txt $O/test_diag.txt 2>/dev/null
timeout 900 bash tools/ab_libs.sh "default NS2VC_CONV_TS=0" "default NS2VC_CONV_TS=1" "default NS2VC_CONV_TS=1 NS2VC_TS_NL=8" "tpf0 NS2VC_CONV_TS=1 NS2VC_TS_NL=8" "default NS2VC_CONV_TS=1 NS2VC_TS_NL=8 NS2VC_TS_BN128_MIN=300" > $O/ab.txt 2>&1
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 --ops $O/ops_ts.txt > $O/bench_ts.json 2> $O/bench_ts.err
timeout 600 python tools/overlap_partition.py > $O/overlap_partition.txt 2>&1
tail -n 5 $O/gpu_tests.txt; cat $O/ab.txt; grep -v amdgpu $O/overlap_partition.txt
