mkdir -p gpurun_out/s13; rm -f gpurun_out/s13/xattn.txt
for v in default xa1 xa2 xa4 xa8 xa15; do
  echo "## $v" >> gpurun_out/s13/xattn.txt
  if [ $v = default ]; then unset NS2VC_LIB; else export NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so; fi
  timeout 200 python tools/xattn_bench.py 2>&1 | grep -v amdgpu >> gpurun_out/s13/xattn.txt
done
cat gpurun_out/s13/xattn.txt
