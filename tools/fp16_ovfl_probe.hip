// Does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 clamp finite overflow to +-65504 on gfx950?
// (and keep +-inf inputs as inf).  hipcc --offload-arch=gfx950 -O3 tools/fp16_ovfl_probe.hip -o tools/bin/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, unsigned* out, int setmode) {
  if (setmode) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1) = 1
  const int i = threadIdx.x;
  f32x2_t v = {in[2 * i], in[2 * i + 1]};
  union { f16x2_t h; unsigned u; } r;
  r.h = __builtin_convertvector(v, f16x2_t);
  out[i] = r.u;
  out[64 + i] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)in[2 * i]);      // scalar v_cvt_f16_f32
}
int main() {
  float h_in[128];
  const float vals[8] = {1.0f, 65504.0f, 65520.0f, 1e6f, -1e6f, INFINITY, -INFINITY, 70000.0f};
  for (int i = 0; i < 128; ++i) h_in[i] = vals[i % 8];
  float* d_in; unsigned* d_out;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 128 * 4);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, mode);
    unsigned h_out[128];
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d  packed:", mode);
    for (int i = 0; i < 4; ++i) printf(" %04x %04x", h_out[i] & 0xffff, h_out[i] >> 16);
    printf("   scalar:");
    for (int i = 0; i < 4; ++i) printf(" %04x", h_out[64 + i] & 0xffff);
    printf("\n");
  }
  printf("inputs: 1 65504 65520 1e6 -1e6 inf -inf 70000  (7bff = 65504, 7c00 = inf)\n");
  return 0;
}
