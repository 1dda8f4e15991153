mkdir -p gpurun_out/s27
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "attention or attn or golden or bench_shape" 2>&1 | tail -4 > gpurun_out/s27/tests.txt; cat gpurun_out/s27/tests.txt
for i in 1 2; do bash tools/ab_libs.sh "head" "default"; done > gpurun_out/s27/ab.txt 2>&1; cat gpurun_out/s27/ab.txt
export NS2VC_DEBUG_ENV=1
python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s27/ops_new.txt > /dev/null 2>&1
NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/head/libns2vc_hip.so python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s27/ops_head.txt > /dev/null 2>&1
grep "attn1.sdpa" gpurun_out/s27/ops_new.txt | cut -f1,3 | head -4; grep "attn1.sdpa" gpurun_out/s27/ops_head.txt | cut -f1,3 | head -4
