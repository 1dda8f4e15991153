#!/bin/bash
# Round-2 GPU session: fused feed-forward kernel -- kernel test, engine parity, same-box A/B
O=gpurun_out; mkdir -p $O; rm -f $O/test_diag.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "ffn or plan_variants or env_switches or bench_shape or golden_2s or deterministic" > $O/r2_pytest_ffn.log 2>&1; echo "pytest rc=$?"; tail -12 $O/r2_pytest_ffn.log
grep -n "ffn fused\|FAIL" $O/test_diag.txt | head -40
ab() { local label=$1; shift; ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
  timeout 300 python bench.py --skip-cpu --skip-fp32 --steps 20 --warmup 20 --reps 5 "$@" 2>> $O/r2_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['ms_per_step'],4), 'min', round(d['timing']['min_ms_per_step'],4), {k:round(v['ms_per_step'],3) for k,v in d['roofline']['families'].items() if k not in ('copy','other')}, d['launches_per_step'])" ); }
for i in 1 2; do
  ab fp16_fused -- --precision fp16
  ab fp16_unfused NS2VC_FUSE_FFN=0 -- --precision fp16
done
python bench.py --skip-cpu --skip-fp32 --ops $O/r2_ops_ffn.txt > /dev/null 2>&1; grep -E "ffn|geglu|ff.out" $O/r2_ops_ffn.txt | head -20
