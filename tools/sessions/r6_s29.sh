mkdir -p gpurun_out/s29
for i in 1 2; do bash tools/ab_libs.sh "default" "default NS2VC_TS_BN128_MIN=110" "default NS2VC_TS_BN128_MIN=200" "default NS2VC_TS_BN128_MIN=240"; done > gpurun_out/s29/ab.txt 2>&1; cat gpurun_out/s29/ab.txt
