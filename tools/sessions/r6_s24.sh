mkdir -p gpurun_out/s24
for i in 1 2; do bash tools/ab_libs.sh "default" "xb4" "xb3" "xb5"; done > gpurun_out/s24/ab.txt 2>&1; cat gpurun_out/s24/ab.txt
