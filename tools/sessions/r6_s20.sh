mkdir -p gpurun_out/s20
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "split_io or hi_lo or solver_update or exact_io or groupnorm_in" 2>&1 | tail -8 > gpurun_out/s20/tests.txt; cat gpurun_out/s20/tests.txt
bash tools/ab_libs.sh "head" "default" "default NS2VC_SPLIT_IO=1" "default NS2VC_FUSE_SOLVER=1" > gpurun_out/s20/ab.txt 2>&1; cat gpurun_out/s20/ab.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/s20/full.txt; cat gpurun_out/s20/full.txt
