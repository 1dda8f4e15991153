mkdir -p gpurun_out/s26
for i in 1 2; do bash tools/ab_libs.sh "default" "pf0" "xb2" "gwt0"; done > gpurun_out/s26/ab.txt 2>&1; cat gpurun_out/s26/ab.txt
