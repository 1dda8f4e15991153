mkdir -p gpurun_out/s22
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/s22/full.txt; cat gpurun_out/s22/full.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s22/smoke.txt 2>&1; tail -5 gpurun_out/s22/smoke.txt
bash tools/sessions/r6_final.sh
