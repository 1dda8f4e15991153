# r6 final evidence set at HEAD: driver command twice, --full once, rocprofv3 stats + family times + HBM + MFMA (pmc_profile.sh), L2 (pmc_l2.sh), determinism probe
mkdir -p gpurun_out/final
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final/bench_driver_cmd_1.json 2> gpurun_out/final/bench_driver_cmd_1.err
cp bench_detail.json gpurun_out/final/bench_detail_driver_cmd.json
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final/bench_driver_cmd_2.json 2> gpurun_out/final/bench_driver_cmd_2.err
( time python bench.py --full --detail-json gpurun_out/final/bench_full_detail.json ) > gpurun_out/final/bench_full_line.json 2> gpurun_out/final/bench_full.err
tail -3 gpurun_out/final/bench_driver_cmd_1.err gpurun_out/final/bench_full.err
bash tools/pmc_profile.sh r06 > gpurun_out/final/pmc_profile.txt 2>&1
bash tools/pmc_l2.sh r06 > gpurun_out/final/pmc_l2.txt 2>&1
timeout 900 python tools/determinism_probe.py --steps 4 --more 21 --forwards 4 > gpurun_out/final/determinism.txt 2>&1
tail -5 gpurun_out/final/determinism.txt
ls gpurun_out/r06_*
