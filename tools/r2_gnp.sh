#!/bin/bash
# GroupNorm-in-producer bring-up: kernel + engine tests, then a same-box A/B of the option
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "groupnorm_in_producer" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_engine_gpu.py -q -x -k "producer or env_switches or plan_variants" 2>&1 | tail -15
for i in 1 2; do
  for v in 1 0; do
    NS2VC_GN_PRODUCER=$v timeout 300 python bench.py --skip-cpu --skip-fp32 --steps 20 --warmup 3 2>> gpurun_out/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gn_producer=$v', round(d['ms_per_step'],4), d['launches_per_step'], 'faults', d['grid_barrier_faults'], 'loop', d['loop_check'], {k:round(v['ms_per_step'],3) for k,v in d['roofline']['families'].items() if k not in ('copy','other')})"
  done
done
