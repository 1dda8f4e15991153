mkdir -p gpurun_out/s21
bash tools/ab_libs.sh "default" "xb8" "xb4" "nld4" > gpurun_out/s21/ab.txt 2>&1; cat gpurun_out/s21/ab.txt
