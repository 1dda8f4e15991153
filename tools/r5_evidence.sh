#!/bin/bash
# round-5 evidence set at one commit, one box: rocprofv3 kernel stats of the bench command + family times, the separate PMC passes (HBM bytes, MFMA / VALU, LDS),
# the determinism probe, the driver's command (full default bench line), the full GPU suite's diagnostics.   gpurun -- 'NS2VC_COMMIT=<hash> bash tools/r5_evidence.sh'
cd "$(dirname "$0")/.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
bash tools/pmc_profile.sh r05 > $O/r05_pmc_profile.log 2>&1
bash tools/pmc_lds.sh r05 >> $O/r05_pmc_profile.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python tools/determinism_probe.py --steps 20 --more 21 --forwards 30 > $O/r05_determinism_probe_25loops.txt 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_driver_cmd.json 2> $O/r05_bench_driver_cmd.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/r05_gpu_tests.txt
cp $O/test_diag.txt $O/test_diag_r05.txt 2>/dev/null
timeout 300 python bench.py --skip-cpu --skip-fp32 --skip-others --skip-strong --steps 20 --warmup 10 --reps 3 --ops $O/r05_ops_per_launch.txt > /dev/null 2>&1
tail -3 $O/r05_gpu_tests.txt; tail -5 $O/r05_pmc_profile.log; tail -4 $O/r05_determinism_probe_25loops.txt; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_driver_cmd.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','launches_per_step','gn_prologue_workgroups_alone','attention_fallback_workgroups')})
print(d['roofline'].get('frac'), d['roofline'].get('frac_rocprof'), d['roofline'].get('frac_isolated'), d['parity'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d.get('bf16_as_stated'))
PY
