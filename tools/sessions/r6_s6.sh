mkdir -p gpurun_out/s6
rm -f gpurun_out/s6/trace.txt
for v in trace_chunk trace_chunk_a1; do
  echo "## $v" >> gpurun_out/s6/trace.txt
  NS2VC_LIB=$PWD/ns2vc_amd/lib/variants/$v/libns2vc_hip.so timeout 300 python tools/ts_trace.py >> gpurun_out/s6/trace.txt 2>&1
done
cat gpurun_out/s6/trace.txt
