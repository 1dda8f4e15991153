mkdir -p gpurun_out/s28
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/s28/full.txt; cat gpurun_out/s28/full.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s28/smoke.txt 2>&1; tail -4 gpurun_out/s28/smoke.txt
bash tools/sessions/r6_final.sh
