mkdir -p gpurun_out/s19
bash tools/ab_libs.sh "head" "default" "nosol" "nosol NS2VC_SPLIT_IO=1" > gpurun_out/s19/ab.txt 2>&1; cat gpurun_out/s19/ab.txt
