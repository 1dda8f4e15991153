mkdir -p gpurun_out/s23
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > gpurun_out/s23/full.txt; cat gpurun_out/s23/full.txt
