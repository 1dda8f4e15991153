#!/bin/bash
# Round-2 GPU session 1: full GPU test suite, default bench line (fp16 + parity + fp32 block + CPU baseline), same-box A/B of
# the plan options and precisions, config-5 shape, multi-rank launch check.  Everything lands in gpurun_out/.
set -u
O=gpurun_out
mkdir -p $O
rm -f $O/test_diag.txt
echo "== pytest -m gpu" ; date
timeout 900 python -m pytest tests -m gpu -x -q > $O/r2_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_pytest.log
tail -5 $O/r2_pytest.log
echo "== bench default" ; date
timeout 600 python bench.py --ops $O/r2_ops_fp16.txt > $O/r2_bench_fp16.json 2> $O/r2_bench_fp16.err; echo "bench rc=$?"
tail -c 600 $O/r2_bench_fp16.json; echo
echo "== A/B (ms/step; families)" ; date
ab() {  # label, env..., -- bench args
  local label=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --skip-cpu --skip-fp32 --steps 20 --warmup 20 --reps 5 "$@" 2>> $O/r2_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), 'min', round(d['timing']['min_ms_per_step'],4), {k:round(v['ms_per_step'],3) for k,v in d['roofline']['families'].items() if k not in ('copy','other')}, d['launches_per_step'])" )
}
for i in 1 2; do
  ab fp16_default -- --precision fp16
  ab fp16_nofold NS2VC_FOLD_FF=0 -- --precision fp16
  ab bf16_default -- --precision bf16
  ab fp16_lnexplicit NS2VC_LN_LINEAR=0 -- --precision fp16
done 2>&1 | tee $O/r2_ab.txt
echo "== config 2 (B=8, 50-step dpm) and config 5 shape (30 s, B=8)"; date
timeout 300 python bench.py --skip-cpu --skip-fp32 --batch 8 --solver dpmsolver++ --steps 50 --warmup 50 --reps 3 > $O/r2_bench_cfg2.json 2>> $O/r2_ab.err
timeout 600 python bench.py --skip-cpu --skip-fp32 --seconds 30 --batch 8 --solver dpmsolver++ --steps 50 --warmup 50 --reps 3 --ops $O/r2_ops_cfg5.txt > $O/r2_bench_cfg5.json 2>> $O/r2_ab.err
python - <<'PY'
import json
for f in ("gpurun_out/r2_bench_cfg2.json", "gpurun_out/r2_bench_cfg5.json"):
    try:
        d = json.load(open(f)); print(f, d["ms_per_step"], d["config"]["workload"], {k: round(v["ms_per_step"], 3) for k, v in d["roofline"]["families"].items()})
    except Exception as e:
        print(f, "FAILED", e)
PY
echo "== multi-rank launch on a 1-GPU box must fail loudly"; date
python bench.py --gpus 2 --skip-cpu --skip-fp32 > $O/r2_gpus2.out 2> $O/r2_gpus2.err; echo "gpus2 rc=$? (expected non-zero)"; tail -2 $O/r2_gpus2.err
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6
date
