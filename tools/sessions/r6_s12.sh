mkdir -p gpurun_out/s12
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "cross_attention or ffn_fused" 2>&1 | tail -4 > gpurun_out/s12/tests.txt; cat gpurun_out/s12/tests.txt
grep "cross-attention inside" gpurun_out/test_diag.txt | tail -6
bash tools/ab_libs.sh "default" "default NS2VC_FUSE_XATTN=0" > gpurun_out/s12/ab.txt 2>&1; cat gpurun_out/s12/ab.txt
export NS2VC_DEBUG_ENV=1
python bench.py --skip-cpu --detail-json= --steps 20 --warmup 10 --reps 3 --ops gpurun_out/s12/ops_xatt.txt > /dev/null 2>&1
grep -E "ffn\[|attn2.sdpa" gpurun_out/s12/ops_xatt.txt | cut -f1,3 | head -12
