#!/bin/bash
# rocprofv3 PMC pass for the LDS pipe (its own run, kernel trace only): bank-conflict cycles vs LDS-active cycles per kernel.
#   gpurun -- 'NS2VC_COMMIT=<hash> bash tools/pmc_lds.sh r02'   -> gpurun_out/<tag>_pmc_lds.json
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/prof_${TAG}_lds \
  -- python $R/bench.py --skip-cpu --detail-json= --steps 4 --warmup 4 --reps 1 > /dev/null 2> /tmp/prof_${TAG}_lds.err
python $R/tools/pmc_lds_summarize.py "$(find /tmp/prof_${TAG}_lds -name '*counter_collection.csv' | head -1)" $R/gpurun_out/${TAG}_pmc_lds.json
