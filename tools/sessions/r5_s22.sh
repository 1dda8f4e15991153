#!/bin/bash
# upper bounds: what would free result stores buy?  (diagnostic builds: stage-2 stores of rowchain_kernel / output stores of attn_kernel removed)
cd "$(dirname "$0")/.."
O=gpurun_out/s22; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/ab_libs.sh "default" "rc_nostore" "attn_nostore" > $O/ab.txt 2>&1
cat $O/ab.txt
