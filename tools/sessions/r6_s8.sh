mkdir -p gpurun_out/s8
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "in_loop or reproducible" 2>&1 | tail -15 > gpurun_out/s8/ktests.txt; cat gpurun_out/s8/ktests.txt
export NS2VC_DEBUG_ENV=1
python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s8/ops_inloop.txt > /dev/null 2>&1
NS2VC_GN_INLOOP=0 python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s8/ops_prologue.txt > /dev/null 2>&1
paste <(grep norm gpurun_out/s8/ops_inloop.txt | cut -f1,3) <(grep norm gpurun_out/s8/ops_prologue.txt | cut -f3)
