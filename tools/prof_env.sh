#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command under an environment switch: bash tools/prof_env.sh TAG VAR=value ...
# -> gpurun_out/<TAG>_kernel_stats.csv (in-situ kernel durations)
export NS2VC_DEBUG_ENV=1   # the plan switches (NS2VC_FUSE_*, NS2VC_CONV_TS, ...) are only read under this (r5)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_$TAG
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --skip-cpu --detail-json= --steps 20 --warmup 20 --reps 3 > /tmp/prof_$TAG.out 2>/tmp/prof_$TAG.err
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
tail -1 /tmp/prof_$TAG.out > $R/gpurun_out/${TAG}_bench.json
