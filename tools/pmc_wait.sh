#!/bin/bash
# rocprofv3 PMC pass for the wave-cycle split (its own run, kernel trace only): parked / issue-stalled / issuing cycles and the VALU mix per kernel.
#   gpurun -- 'NS2VC_COMMIT=<hash> bash tools/pmc_wait.sh r05'   -> gpurun_out/<tag>_pmc_wave_cycles.json
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_wait /tmp/prof_${TAG}_wait2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d /tmp/prof_${TAG}_wait \
  -- python $R/bench.py --skip-cpu --detail-json= --steps 4 --warmup 4 --reps 1 > /dev/null 2> /tmp/prof_${TAG}_wait.err
f=$(find /tmp/prof_${TAG}_wait -name '*counter_collection.csv' | head -1)
if [ -z "$f" ]; then   # (a counter of the list not available on this build of rocprofv3: retry without the transcendental count)
  tail -3 /tmp/prof_${TAG}_wait.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/prof_${TAG}_wait2 \
    -- python $R/bench.py --skip-cpu --detail-json= --steps 4 --warmup 4 --reps 1 > /dev/null 2> /tmp/prof_${TAG}_wait.err
  f=$(find /tmp/prof_${TAG}_wait2 -name '*counter_collection.csv' | head -1)
fi
python $R/tools/pmc_wait_summarize.py "$f" $R/gpurun_out/${TAG}_pmc_wave_cycles.json
