// Shared device/host definitions for the NS2VC denoiser engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ns2vc_hip.h"

namespace ns2vc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 2-byte storage type for bf16 activations / weights
struct bf16_t { uint16_t v; };

__host__ __device__ inline uint16_t f32_to_bf16_bits(float f) {
  union { float f; uint32_t u; } x;
  x.f = f;
  uint32_t u = x.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
  return (uint16_t)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_f32(uint16_t h) {
  union { float f; uint32_t u; } x;
  x.u = (uint32_t)h << 16;
  return x.f;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float v) { return v * fast_rcp(1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------
// implicit-GEMM (conv1d k3/k1, linear) arguments
// rows   m = b*Tout + t          (activations, channels-last [B][T][C])
// K idx  k = tap*(c0+c1) + c     (c < c0 -> source 0, else source 1: no-copy concat)
// out[m][n] = epi( sum_k pro(A[m,k]) * W[n][k] )
// ---------------------------------------------------------------------------
enum { TMODE_SAME = 0, TMODE_DOWN2 = 1, TMODE_UP2 = 2 };
enum { PRO_NONE = 0, PRO_BC = 1, PRO_ROW = 2 };

typedef ::ns2vc_gemm_args GemmArgs;   // public POD, include/ns2vc_hip.h
typedef ::ns2vc_attn_args AttnArgs;

enum Precision { PREC_F32 = 0, PREC_BF16 = 1 };

// launchers (defined in the .hip files); return hipError_t
hipError_t launch_gemm(const GemmArgs& g, int prec, hipStream_t s);
hipError_t launch_attention(const AttnArgs& a, int head_dim, int prec, hipStream_t s);
hipError_t init_gemm_attributes();
void set_forced_gemm_tile(int bm, int bn);
hipError_t init_attn_attributes();

// misc kernels (misc.hip)
hipError_t launch_gn_partial(const float* a0, int lda0, int c0, const float* a1, int lda1, int c1,
                             int B, int T, int G, double* partial, int nchunk, int rows_per_chunk, hipStream_t s);
hipError_t launch_gn_coef(const double* partial, int nchunk, int B, int T, int C, int G, float eps,
                          const float* gamma, const float* beta, const float* temb, int ldtemb, int temb_off, int cout,
                          float* pscale, float* pshift, hipStream_t s);
hipError_t launch_ln_stats(const float* x, int ldx, int M, int C, float eps, float* rstats, hipStream_t s);
hipError_t launch_time_embed(const float* t_ptr, int t_stride, const int* step_ptr, int coef_stride,
                             const float* w1t, const float* b1, const float* w2t, const float* b2,
                             const float* aug, float* emb, float* emb_act, int B, int tdim, int edim, hipStream_t s);
hipError_t launch_nct_to_btc(const float* src, int C, int T, int B, float* dst, int ldd, int cpad, hipStream_t s);
hipError_t launch_btc_to_nct(const float* src, int lds, int C, int T, int B, float* dst, hipStream_t s);
hipError_t launch_mask_bias(const uint8_t* mask, int n, float* bias, hipStream_t s);
hipError_t launch_ln_apply(const float* x, int M, int C, float eps, const float* gamma, const float* beta,
                           float* out, int L, int Lout_stride_rows, hipStream_t s);
hipError_t launch_pool_cls(float* seq, int B, int L, int C, const float* pos, hipStream_t s);
hipError_t launch_pool_attn(const float* qkv, int B, int L1, int C, int heads, float* pooled, hipStream_t s);
hipError_t launch_pool_proj(const float* pooled, int B, int C, const float* wt, const float* b, int E,
                            const float* gamma, const float* beta, float eps, float* out, hipStream_t s);
hipError_t launch_solver_update(const float* coef, const int* step_ptr, int ncoef, const float* x0,
                                float* xe, float* xbar, float* d1, float* mprev, size_t n, hipStream_t s);
hipError_t launch_step_advance(int* step_ptr, hipStream_t s);
hipError_t launch_fill_i32(int* p, int v, hipStream_t s);

}  // namespace ns2vc
