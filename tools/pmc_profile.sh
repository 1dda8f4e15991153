#!/bin/bash
# rocprofv3 evidence for profiles/: kernel trace + stats, then HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and
# MFMA / VALU activity, each counter set in its OWN run (never combined with sys/hip traces).  GPU box only:
#   gpurun -- 'bash tools/pmc_profile.sh r02'            -> gpurun_out/<tag>_*.{csv,json}
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --skip-cpu --detail-json= --steps 4 --warmup 4 --reps 1"
rm -rf /tmp/prof_$TAG*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_stats -- python $R/bench.py --skip-cpu --steps 20 --warmup 20 --reps 3 --detail-json $O/${TAG}_stats_bench.json > $O/${TAG}_stats_line.json 2> /tmp/prof_${TAG}_stats.err
cp "$(find /tmp/prof_${TAG}_stats -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats_bench_steps20.csv
python $R/tools/family_times.py $TAG $O/${TAG}_kernel_stats_bench_steps20.csv $O/${TAG}_stats_bench.json $O
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_${TAG}_$C -- $CMD > /dev/null 2> /tmp/prof_${TAG}_$C.err
  cp "$(find /tmp/prof_${TAG}_$C -name '*counter_collection.csv' | head -1)" /tmp/${TAG}_$C.csv
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv \
  -d /tmp/prof_${TAG}_mfma -- $CMD > /dev/null 2> /tmp/prof_${TAG}_mfma.err
cp "$(find /tmp/prof_${TAG}_mfma -name '*counter_collection.csv' | head -1)" /tmp/${TAG}_mfma.csv
python $R/tools/pmc_summarize.py $TAG /tmp/${TAG}_FETCH_SIZE.csv /tmp/${TAG}_WRITE_SIZE.csv /tmp/${TAG}_mfma.csv $O
ls -la $O/${TAG}_*
