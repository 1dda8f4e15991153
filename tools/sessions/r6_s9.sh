mkdir -p gpurun_out/s9
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "in_loop or groupnorm_prologue or tapshare or fused_shortcut" 2>&1 | tail -3 > gpurun_out/s9/ktests.txt; cat gpurun_out/s9/ktests.txt
bash tools/ab_libs.sh "default" "nld8" "tap" > gpurun_out/s9/ab.txt 2>&1; cat gpurun_out/s9/ab.txt
