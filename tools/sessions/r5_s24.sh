#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s24; mkdir -p $O
export TMPDIR=/tmp
timeout 900 bash tools/ab_libs.sh "prev" "default" "rc_wt" "rc_direct_wt" > $O/ab.txt 2>&1
cat $O/ab.txt
