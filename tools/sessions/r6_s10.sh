mkdir -p gpurun_out/s10
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "solver_update_in_conv_out or inside_the_conv_loop or sampler_golden" 2>&1 | tail -8 > gpurun_out/s10/tests.txt; cat gpurun_out/s10/tests.txt
bash tools/ab_libs.sh "default" "default NS2VC_FUSE_SOLVER=0" > gpurun_out/s10/ab.txt 2>&1; cat gpurun_out/s10/ab.txt
