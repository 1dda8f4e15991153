mkdir -p gpurun_out/s15
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "sampler or solver_update or call_order or mixed_precision" 2>&1 | tail -4 > gpurun_out/s15/tests.txt; cat gpurun_out/s15/tests.txt
bash tools/ab_libs.sh "default" "default NS2VC_FORK_TEMB=0" > gpurun_out/s15/ab.txt 2>&1; cat gpurun_out/s15/ab.txt
