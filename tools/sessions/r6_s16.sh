mkdir -p gpurun_out/s16
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "exact_io or golden" 2>&1 | tail -6 > gpurun_out/s16/tests.txt; cat gpurun_out/s16/tests.txt
grep "exact_io" gpurun_out/test_diag.txt | tail -2
bash tools/ab_libs.sh "default" "default NS2VC_EXACT_IO=1" > gpurun_out/s16/ab.txt 2>&1; cat gpurun_out/s16/ab.txt
export NS2VC_DEBUG_ENV=1
NS2VC_EXACT_IO=1 python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s16/ops_xio.txt > /dev/null 2>&1
grep -E "^time_emb|^conv_in|^conv_out|^time_embed" gpurun_out/s16/ops_xio.txt | cut -f1,3
