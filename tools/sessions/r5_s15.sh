#!/bin/bash
# r5 session 15: token-stationary GEGLU projection (geglu.hip): kernel + engine tests, drop-in late re-check, same-box A/B, per-launch times
cd "$(dirname "$0")/.."
O=gpurun_out/s15; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "geglu_token_stationary" 2>&1 | tail -25 > $O/kernel_tests.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "token_stationary or block_by_block or golden or odd_shapes or plan_variants" 2>&1 | tail -12 > $O/engine_tests.txt
timeout 600 python -m pytest tests/test_dropin_gpu.py -x -q 2>&1 | tail -12 > $O/dropin_tests.txt
timeout 600 bash tools/ab_libs.sh "default NS2VC_FUSE_GEGLU=0" "default NS2VC_FUSE_GEGLU=1" > $O/ab.txt 2>&1
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 10 --warmup 5 --reps 1 --ops $O/ops.tsv > $O/bench.txt 2>&1
grep -h "geglu\|GEGLU" gpurun_out/test_diag.txt | tail -20
tail -n 6 $O/kernel_tests.txt $O/engine_tests.txt $O/dropin_tests.txt; cat $O/ab.txt; grep "geglu" $O/ops.tsv
