mkdir -p gpurun_out/s17
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "split_io or hi_lo or golden or solver_update_in_conv_out or exact_io" 2>&1 | tail -8 > gpurun_out/s17/tests.txt; cat gpurun_out/s17/tests.txt
grep -E "split_io|hi \+ lo" gpurun_out/test_diag.txt | tail -12
bash tools/ab_libs.sh "head" "default" "default NS2VC_SPLIT_IO=1" > gpurun_out/s17/ab.txt 2>&1; cat gpurun_out/s17/ab.txt
export NS2VC_DEBUG_ENV=1
NS2VC_SPLIT_IO=1 python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s17/ops_pio.txt > /dev/null 2>&1
python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s17/ops_def.txt > /dev/null 2>&1
grep -E "^conv_in|^conv_out|^solver" gpurun_out/s17/ops_pio.txt | cut -f1,3
grep -E "^conv_in|^conv_out|^solver" gpurun_out/s17/ops_def.txt | cut -f1,3
