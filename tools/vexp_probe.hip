// Issue cost of the softmax's VALU instructions next to the MFMA pipe on gfx950 ("measure before
// believing VALU-bound").  One workgroup on one CU; every wave runs a long unrolled loop of independent instructions and
// times it with s_memtime (shader clock) and s_memrealtime (100 MHz), so cycles per wave64 instruction come out directly.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/vexp_probe tools/vexp_probe.hip && /tmp/vexp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Mode { EXP = 0, MAX, CVT, PKFMA, MFMA, MFMA_EXP8, MFMA_EXP16, FMA, MAX3, EXP_MAX_CVT, NMODE };
static const char* kName[NMODE] = {"v_exp_f32", "v_max_f32", "v_cvt_pk_f16_f32 (v_cvt_pkrtz)", "v_pk_fma_f32 (2 fp32 / lane)",
                                   "v_mfma_f32_32x32x16_f16", "[1 MFMA + 8 v_exp_f32] x4 (same wave)", "[1 MFMA + 16 v_exp_f32] x4 (same wave)",
                                   "v_fma_f32", "v_max3_f32", "[1 MFMA + 8 exp + 8 max + 4 cvt_pk] x4 (the attention tile mix)"};
static const int kInstr[NMODE] = {16, 16, 16, 16, 4, 36, 68, 16, 16, 84};     // instructions per loop body

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* sink, unsigned long long* out, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = -0.001f * (float)(threadIdx.x + i + 1);
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (float)threadIdx.x); b[i] = (_Float16)1.0f; }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == EXP) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if constexpr (MODE == MAX) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
    } else if constexpr (MODE == MAX3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
    } else if constexpr (MODE == FMA) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
    } else if constexpr (MODE == CVT) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
    } else if constexpr (MODE == PKFMA) {
      f32x2* p = reinterpret_cast<f32x2*>(v);
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    } else if constexpr (MODE == MFMA) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    } else if constexpr (MODE == MFMA_EXP8 || MODE == MFMA_EXP16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {                       // (static accumulator index: 4 bodies per iteration)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < (MODE == MFMA_EXP8 ? 8 : 16); ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      }
    } else if constexpr (MODE == EXP_MAX_CVT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[8 + i]) : "v"(v[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(v[12 + i]) : "v"(v[2 * i]), "v"(v[2 * i + 1]));
      }
    }
  }
  asm volatile("s_nop 0" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) { out[2 * (threadIdx.x >> 6)] = t1 - t0; out[2 * (threadIdx.x >> 6) + 1] = r1 - r0; }
}

template <int MODE> static void run(float* sink, unsigned long long* dout, int waves_per_simd) {
  const int iters = 4000, threads = 256 * waves_per_simd;
  std::vector<unsigned long long> h(2 * 16);
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, sink, dout, 100);
  hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(threads), 0, 0, sink, dout, iters);
  hipDeviceSynchronize();
  hipMemcpy(h.data(), dout, sizeof(unsigned long long) * 2 * (threads / 64), hipMemcpyDeviceToHost);
  double cyc = 0, ref = 0;
  for (int w = 0; w < threads / 64; ++w) { cyc += (double)h[2 * w]; ref += (double)h[2 * w + 1]; }
  cyc /= threads / 64; ref /= threads / 64;
  const double per = cyc / ((double)iters * kInstr[MODE]);
  printf("%-46s waves/SIMD %d: %8.2f shader-clock cycles per instruction per wave  (%7.1f per loop body; clock %.0f MHz; SIMD-time per instr %.2f)\n",
         kName[MODE], waves_per_simd, per, cyc / iters, cyc / (ref / 100.0), per / waves_per_simd);
}

int main() {
  float* sink; unsigned long long* dout;
  hipMalloc(&sink, 64); hipMalloc(&dout, 4096);
  for (int w = 1; w <= 2; ++w) {
    run<EXP>(sink, dout, w); run<MAX>(sink, dout, w); run<MAX3>(sink, dout, w); run<FMA>(sink, dout, w); run<CVT>(sink, dout, w); run<PKFMA>(sink, dout, w);
    run<MFMA>(sink, dout, w); run<MFMA_EXP8>(sink, dout, w); run<MFMA_EXP16>(sink, dout, w); run<EXP_MAX_CVT>(sink, dout, w);
  }
  return 0;
}
