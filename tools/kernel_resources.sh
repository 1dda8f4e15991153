#!/bin/bash
# register / scratch audit of every kernel of the library (no GPU needed: hipcc's kernel-resource-usage remarks):  bash tools/kernel_resources.sh [out.txt]
# r6: a solver epilogue behind a run-time test had put 26 spilled VGPRs (108 B of scratch per lane) into EVERY 128-column tile of the tap-sharing conv kernel;
# a scratch reload is a vmcnt(0) drain next to hand-counted LDS-DMA.  The default path must list no kernel with scratch.
cd "$(dirname "$0")/../ns2vc_amd/csrc"
OUT=${1:-../../profiles/r06_kernel_resources.txt}
{
echo "# hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage, every kernel of libns2vc_hip.so: VGPRs | AGPRs | scratch B/lane | waves/SIMD | spilled VGPRs"
for f in gemm convts ffn geglu rowchain attn misc; do
  flags=""; [ $f = attn ] && flags="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include $flags -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kres_$$.o 2>&1 |
    grep -E "Function Name|    VGPRs:|AGPRs:|ScratchSize|Occupancy|VGPRs Spill" | sed 's/.*remark: *//; s/ *\[-Rpass.*//; s/Function Name: //; s/.*: //' | paste - - - - - - |
    while IFS=$'\t' read -r name v a sc occ sp; do printf "%-8s %4s %4s %5s %2s %3s  %s\n" $f "$v" "$a" "$sc" "$occ" "$sp" "$(echo "$name" | c++filt | cut -c1-150)"; done
done
rm -f /tmp/kres_$$.o
} > "$OUT"
echo "kernels: $(grep -vc '^#' "$OUT"); with scratch: $(grep -v '^#' "$OUT" | awk '$4 != 0' | wc -l)"
grep -v '^#' "$OUT" | awk '$4 != 0' | cut -c1-200
