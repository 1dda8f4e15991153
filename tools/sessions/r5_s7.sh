#!/bin/bash
# round-5 GPU session 7: CU partition of one MI355X between the denoiser and the PyTorch stages; corrected per-launch table
cd "$(dirname "$0")/.."
O=gpurun_out/s7; mkdir -p $O
export TMPDIR=/tmp
timeout 420 python -u tools/overlap_partition.py > $O/overlap_partition.txt 2>&1
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 --ops $O/ops.txt > $O/bench.json 2> $O/bench.err
grep -v amdgpu $O/overlap_partition.txt; tail -c 600 $O/bench.json
