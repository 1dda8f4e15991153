#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (in-situ kernel durations), summary CSV into gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_r2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r2 -- python $R/bench.py --skip-cpu --detail-json= --steps 20 --warmup 20 --reps 3 > /tmp/prof_r2.out 2>/tmp/prof_r2.err
f=$(find /tmp/prof_r2 -name "*kernel_stats.csv" | head -1)
echo "stats file: $f"
cp "$f" $R/gpurun_out/r2_kernel_stats.csv
head -25 "$f" | cut -c1-200
tail -2 /tmp/prof_r2.out | cut -c1-300
