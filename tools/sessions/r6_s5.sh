mkdir -p gpurun_out/s5; rm -f gpurun_out/s5/ablate.txt
for v in default tap chunk_a1 chunk_a2; do
  echo "## $v" >> gpurun_out/s5/ablate.txt
  if [ $v = default ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
  timeout 300 python tools/gemm_sweep.py --ts --rotate 8 2>&1 | grep -E "conv3|configs" >> gpurun_out/s5/ablate.txt
done
unset NS2VC_LIB
cat gpurun_out/s5/ablate.txt
bash tools/ab_libs.sh "default" "tap" > gpurun_out/s5/ab_chunk.txt 2>&1; cat gpurun_out/s5/ab_chunk.txt
