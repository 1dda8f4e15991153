#!/bin/bash
# r6: L2 (TCC) hit rate and memory-side request mix per kernel of the bench step -- what the K loops are actually bound by.
# Each counter group is its OWN rocprofv3 pass (4 TCC slots per pass on gfx950), --kernel-trace only, never combined with other traces.
#   gpurun -- 'bash tools/pmc_l2.sh r06'            -> gpurun_out/<tag>_pmc_l2.json
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --skip-cpu --steps 4 --warmup 4 --reps 1 --detail-json= ${PMC_BENCH_ARGS}"
i=0
FILES=""
for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_WRITE_sum" \
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum TCC_STREAMING_REQ_sum" \
         "TCC_NORMAL_EVICT_sum TCC_NORMAL_WRITEBACK_sum TCC_BUSY_sum TCC_CYCLE_sum" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
         "TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_TCC_RW_READ_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/prof_${TAG}_l2_$i
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d /tmp/prof_${TAG}_l2_$i -- $CMD > /dev/null 2> /tmp/prof_${TAG}_l2_$i.err
  F="$(find /tmp/prof_${TAG}_l2_$i -name '*counter_collection.csv' | head -1)"
  if [ -n "$F" ]; then cp "$F" /tmp/${TAG}_l2_$i.csv; FILES="$FILES /tmp/${TAG}_l2_$i.csv"; else echo "pass $i ($G) produced no counters:"; tail -3 /tmp/prof_${TAG}_l2_$i.err; fi
done
python $R/tools/pmc_l2_summarize.py $TAG $O $FILES
