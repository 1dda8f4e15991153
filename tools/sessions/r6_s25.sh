mkdir -p gpurun_out/s25
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/s25/full.txt; cat gpurun_out/s25/full.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
