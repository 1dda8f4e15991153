// Shared device/host definitions for the NS2VC denoiser engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "../../include/ns2vc_hip.h"

namespace ns2vc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 2-byte storage types of the 16-bit operand precisions (activations / weights as the MFMA reads them)
struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

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

// IEEE binary16, round to nearest even; overflow -> inf, subnormals kept (host side: weight packing, test helpers)
__host__ __device__ inline uint16_t f32_to_f16_bits(float f) {
  union { float f; uint32_t u; } x;
  x.f = f;
  const uint32_t sign = (x.u >> 16) & 0x8000u;
  uint32_t a = x.u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);             // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);            // >= 65520 rounds to inf
  if (a < 0x38800000u) {                                              // below the smallest normal (2^-14): subnormal / zero
    if (a < 0x33000000u) return (uint16_t)sign;                       // < 2^-25 -> 0
    const int e = (int)(a >> 23);                                     // biased fp32 exponent, 102..112
    const uint32_t m = (a & 0x7fffffu) | 0x800000u;                   // 24-bit significand
    const int shift = 126 - e;                                        // 14..24: bits dropped
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) ++r;
    return (uint16_t)(sign | r);
  }
  a += 0xc8000000u;                                                   // rebias exponent 127 -> 15 (subtract 112 << 23)
  a += 0xfffu + ((a >> 13) & 1u);                                     // round to nearest even on the 13 dropped bits
  return (uint16_t)(sign | (a >> 13));
}
__host__ __device__ inline float f16_bits_to_f32(uint16_t h) {
  union { float f; uint32_t u; } x;
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  if (e == 0) {
    if (m == 0) { x.u = sign; return x.f; }
    x.f = (float)m * 5.9604644775390625e-8f;                          // m * 2^-24
    x.u |= sign;
    return x.f;
  }
  if (e == 31) { x.u = sign | 0x7f800000u | (m << 13); return x.f; }
  x.u = sign | ((e + 112u) << 23) | (m << 13);
  return x.f;
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// hardware round-to-nearest-even pack (v_cvt_pk_bf16_f32): lo -> bits [15:0], hi -> bits [31:16]
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  union { bf16x2_t b; uint32_t u; } r;
  r.b = __builtin_convertvector(v, bf16x2_t);
  return r.u;
}
__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// fp16 (IEEE binary16, round to nearest even; v_cvt_pk_f16_f32 on gfx950).  Overflow: every kernel that produces fp16
// operands starts with op_mode_init<f16_t>(), which sets MODE.FP16_OVFL -- finite values beyond +-65504 then convert to
// +-65504 instead of +-inf (an activation outlier saturates instead of turning whole rows into inf/NaN downstream), while
// +-inf inputs stay inf (the attention mask's -inf for tail keys relies on that).  Checked on MI355X by
// tools/fp16_ovfl_probe.hip; costs nothing per element (an explicit v_med3 clamp cost 1.7 % of the step).
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  union { f16x2_t h; uint32_t u; } r;
  r.h = __builtin_convertvector(v, f16x2_t);
  return r.u;
}
__device__ __forceinline__ float f16_lo(uint32_t u) { union { uint32_t u; f16x2_t h; } r; r.u = u; return (float)r.h[0]; }
__device__ __forceinline__ float f16_hi(uint32_t u) { union { uint32_t u; f16x2_t h; } r; r.u = u; return (float)r.h[1]; }

// one interface over the two 16-bit operand types (TM = bf16_t | f16_t): two floats <-> one packed dword
template <typename TM> struct Op16;
template <> struct Op16<bf16_t> {
  __device__ static __forceinline__ uint32_t pack(float lo, float hi) { return pack_bf16x2(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t u) { return bf16_lo(u); }
  __device__ static __forceinline__ float hi(uint32_t u) { return bf16_hi(u); }
};
template <> struct Op16<f16_t> {
  __device__ static __forceinline__ uint32_t pack(float lo, float hi) { return pack_f16x2(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t u) { return f16_lo(u); }
  __device__ static __forceinline__ float hi(uint32_t u) { return f16_hi(u); }
};
// per-kernel setup of the operand type: fp16 -> MODE.FP16_OVFL = 1 (hwreg(HW_REG_MODE, offset 23, size 1)), see above
template <typename TM> __device__ __forceinline__ void op_mode_init() {}
template <> __device__ __forceinline__ void op_mode_init<f16_t>() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

// operand-typed scalar / 4-vector stores and loads (TM = float or bf16_t)
template <typename TM> __device__ __forceinline__ void store_op(TM* p, float v);
template <> __device__ __forceinline__ void store_op<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_op<bf16_t>(bf16_t* p, float v) { p->v = (uint16_t)pack_bf16x2(v, 0.f); }
template <> __device__ __forceinline__ void store_op<f16_t>(f16_t* p, float v) { p->v = (uint16_t)pack_f16x2(v, 0.f); }
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope memory fence: once a kernel has
// issued global stores the compiler puts `s_waitcnt vmcnt(0)` in front of the barrier, i.e. every wave sits through the
// write latency of everything stored so far (1-2 us per barrier in the epilogues, found in the ISA).  Where the barrier only
// protects an LDS staging buffer this is all that is needed.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
// what the operand type drops of v: v - float(TM(v)) -- the `lo` plane of a hi + lo operand pair (r6 split_io: conv_in / conv_out with
// x*w ~= hi(x)*hi(w) + lo(x)*hi(w) + hi(x)*lo(w) on the 16-bit MFMA, 2^-22-ish relative instead of 2^-11)
template <typename TM> __device__ __forceinline__ float op_rest(float v) { return v - Op16<TM>::lo(Op16<TM>::pack(v, 0.f)); }
template <> __device__ __forceinline__ float op_rest<float>(float) { return 0.f; }
template <typename TM> __device__ __forceinline__ void store_op4(TM* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_op4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store_op4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
template <> __device__ __forceinline__ void store_op4<f16_t>(f16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16x2(a, b), pack_f16x2(c, d));
}
// Result stores of the big producers (GEMM epilogues, norm outputs).  NS2VC_WT_STORES=1 (default) issues them
// write-through (sc1): the bytes leave during the kernel instead of as an L2 write-back at the kernel boundary
// (every kernel used to leave 8-30 MB dirty).  Same-box A/B: 4.65 -> 4.45 ms/step; the attention output is better
// left to plain stores (+5 % attention time with sc1), so attn.hip does not use these.
#ifndef NS2VC_WT_STORES
#define NS2VC_WT_STORES 1
#endif
#ifndef NS2VC_WT_MODE
#define NS2VC_WT_MODE 1
#endif
#ifndef NS2VC_WT_MOD            // cache-policy bits of those stores; NS2VC_WT_MODE selects one for A/B builds (1 = shipped).  r4, same box, ms/step:
                                // sc1 3.709 | sc1 nt 3.827 | nt 3.808 | sc0 sc1 3.704 (profiles/r04_ab_store_policy.txt)
#if NS2VC_WT_MODE == 2
#define NS2VC_WT_MOD "sc1 nt"
#elif NS2VC_WT_MODE == 3
#define NS2VC_WT_MOD "nt"
#elif NS2VC_WT_MODE == 4
#define NS2VC_WT_MOD "sc0 sc1"
#else
#define NS2VC_WT_MOD "sc1"
#endif
#endif
__device__ __forceinline__ void out_store16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
#if NS2VC_WT_STORES
  const u32x4_t v = {a, b, c, d};
  asm volatile("global_store_dwordx4 %0, %1, off " NS2VC_WT_MOD "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<uint4*>(p) = make_uint4(a, b, c, d);
#endif
}
__device__ __forceinline__ void out_store8(void* p, uint32_t a, uint32_t b) {
#if NS2VC_WT_STORES
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off " NS2VC_WT_MOD "\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
#endif
}
__device__ __forceinline__ void out_f4(float* p, float a, float b, float c, float d) {
  out_store16(p, __float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d));
}
template <typename TM> __device__ __forceinline__ void out_op4(TM* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void out_op4<float>(float* p, float a, float b, float c, float d) { out_f4(p, a, b, c, d); }
template <> __device__ __forceinline__ void out_op4<bf16_t>(bf16_t* p, float a, float b, float c, float d) { out_store8(p, pack_bf16x2(a, b), pack_bf16x2(c, d)); }
template <> __device__ __forceinline__ void out_op4<f16_t>(f16_t* p, float a, float b, float c, float d) { out_store8(p, pack_f16x2(a, b), pack_f16x2(c, d)); }
template <typename TM> __device__ __forceinline__ void store_op2(TM* p, float a, float b);
template <> __device__ __forceinline__ void store_op2<float>(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
template <> __device__ __forceinline__ void store_op2<bf16_t>(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf16x2(a, b); }
template <> __device__ __forceinline__ void store_op2<f16_t>(f16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_f16x2(a, b); }
#endif

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float silu_f(float v) { return v * fast_rcp(1.0f + __expf(-v)); }
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, branch-free): the exact-GELU of the reference
// (attention.py:295, F.gelu default) to ~1e-7, far inside the 1e-3 parity budget, at ~15 VALU ops.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = fast_rcp(1.0f + 0.3275911f * ax);
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(e, x);
}
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erf_as(v * 0.70710678118654752440f)); }

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

enum Precision { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };
inline size_t operand_bytes(int prec) { return prec == PREC_F32 ? 4 : 2; }
// host-side rounding of a value to the operand type (weight packing, row sums of the rounded weights)
inline uint16_t f32_to_op16_bits(float v, int prec) { return prec == PREC_BF16 ? f32_to_bf16_bits(v) : f32_to_f16_bits(v); }
inline float op16_bits_to_f32(uint16_t b, int prec) { return prec == PREC_BF16 ? bf16_bits_to_f32(b) : f16_bits_to_f32(b); }
// fixed-point scales of the epilogue GroupNorm statistics (order-independent int64 atomics => deterministic)
// One element of the fused solver update (ns2vc_amd/schedule.py documents the recurrence; sampler/dpm_solver.py:433-442, 547-580, sampler/uni_pc.py:471-588):
// the x0 -> eps -> x0 round trip of the reference's model wrapper, literally, then the unified multistep update.  ONE definition, used by
// solver_update_kernel (misc.hip) and by the conv_out epilogue of conv3ts_kernel (convts.hip): the two forms give the same bits.
struct SolverCoef { float alpha, sigma, g0, g1, A, Bc, d1c, pc; };
__device__ __forceinline__ SolverCoef solver_coef(const float* __restrict__ c) { return SolverCoef{c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8]}; }
__device__ __forceinline__ void solver_upd(const SolverCoef& k, float vx0, float vxe, float vxb, float vd1, float vmp, float& oxe, float& oxb, float& od1, float& om) {
  const float eps = (vxe - k.alpha * vx0) / k.sigma;
  const float m = (vxe - k.sigma * eps) / k.alpha;
  const float x = vxb - k.g0 * vd1 - k.g1 * (m - vmp);
  const float nb = k.A * x - k.Bc * m;
  const float nd = k.d1c * (vmp - m);
  oxb = nb; od1 = nd; oxe = nb - k.pc * nd; om = m;
}

constexpr double GN_SUM_SCALE = 268435456.0;   // 2^28
constexpr double GN_SQ_SCALE = 65536.0;        // 2^16

// launchers (defined in the .hip files); return hipError_t
hipError_t launch_gemm(const GemmArgs& g, int prec, hipStream_t s);
size_t xattn_pack_bytes(int B, int Lk, int hd);                            // ffn.hip: the k | v fragment image of the in-kernel cross-attention
hipError_t launch_xattn_pack(const void* k, int ldk, const void* v, int ldv, int B, int Lk, int hd, void* out, int prec, hipStream_t s);
bool gemm_uses_convts(const GemmArgs& g, int prec);                        // would launch_gemm run this launch on the tap-sharing conv kernel (convts.hip)?
int last_gemm_refusal_line();                                               // gemm.hip line of the argument check that refused the last launch on this thread (0: none), cleared by the call
hipError_t launch_attention(const AttnArgs& a, int head_dim, int prec, hipStream_t s);
hipError_t init_gemm_attributes();
// tap-sharing k = 3 conv kernel (convts.hip)
bool convts_eligible(const GemmArgs& g, int prec);
int convts_default_bn(const GemmArgs& g);                                  // g.conv_bn if set, else the heuristic with the process default (debug hooks)
int convts_bn_for(const GemmArgs& g, int bn128_min);                       // the heuristic: 128-column tiles while they still give bn128_min workgroups
void set_convts_bn128_min(int wgs);
int convts_row_blocks(const GemmArgs& g);
hipError_t launch_convts(const GemmArgs& g, int prec, int bn, int nl, int ks, hipStream_t s);
hipError_t init_convts_attributes();
hipError_t pack_conv3_tiled(const float* rows, int N, int ctot, int c2, int prec, std::vector<unsigned char>& out);   // tile-major k = 3 conv weights (convts.hip)
void set_forced_gemm_tile(int bm, int bn, int stages);
void set_gemm_trace(unsigned long long* p);
hipError_t init_attn_attributes();
void set_forced_attn_keys(int keys);            // test hook: 0 = heuristic, 64 / 128 = K/V tile size where the kernel exists
// fused feed-forward + proj_out (ffn.hip), 16-bit operand types only
hipError_t launch_ffn(const ::ns2vc_ffn_args& a, int prec, hipStream_t s);
bool ffn_eligible(int dim, int T, int prec);
hipError_t init_ffn_attributes();

// misc kernels (misc.hip).  "op" buffers are operand-typed (bf16 / fp16 / fp32 by `prec`)
hipError_t launch_gn_partial(const float* a0, int lda0, int c0, const float* a1, int lda1, int c1,
                             int B, int T, int G, double* partial, int nchunk, int rows_per_chunk, hipStream_t s);
hipError_t launch_gn_apply(const float* a0, int lda0, int c0, const float* a1, int lda1, int c1, int B, int T, int G, float eps,
                           const double* partial, int nchunk, const long long* st0, const long long* st1, const float* gamma,
                           const float* beta, const float* temb, int ldtemb, int temb_off, int silu, void* out_op, void* raw_op,
                           int prec, hipStream_t s, int pair = 0);      // pair (16-bit): out_op rows are [hi(C) | lo(C)] (op_rest)
hipError_t launch_ln_apply_op(const float* x, int ldx, int M, int C, float eps, void* out_op, int prec, hipStream_t s);
hipError_t launch_cast_op(const float* x, size_t n, void* out_op, int prec, hipStream_t s, int split = 0);   // split: as launch_solver_update
hipError_t launch_time_embed(const float* t_ptr, int t_stride, const int* step_ptr, int coef_stride,
                             const float* w1t, const float* b1, const float* w2t, const float* b2,
                             const float* aug, float* emb, void* emb_act_op, int prec, int B, int tdim, int edim, hipStream_t s);
hipError_t launch_rowchain(const ::ns2vc_rowchain_args& a, int prec, hipStream_t s);            // rowchain.hip
bool rowchain_eligible(int dim, int n2, int T, int prec);
hipError_t init_rowchain_attributes();
void set_forced_rowchain_tokens(int nt);        // test hook: 0 = heuristic, 1 = 64-token blocks, 2 = 128-token blocks (dim 128 only)
hipError_t launch_emb_from_table(const float* table, const int* step_ptr, const float* aug, float* emb, void* emb_act_op, int prec, int B,
                                 int edim, hipStream_t s);
int probe_xcd_round_robin(unsigned* map8 = nullptr);
//                                             // misc.hip: 1 = workgroup ids 8 apart share an XCD (what gnp_sync relies on), 0 = not, -1 = probe failed
hipError_t launch_zero(void* p, size_t bytes, hipStream_t s, int* counter = nullptr);                                  // bytes: any; p 16-byte aligned
hipError_t launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t s);           // bytes % 16 == 0
hipError_t launch_poison(unsigned pattern, int lds_bytes, unsigned* sink, hipStream_t s);
hipError_t launch_nct_to_btc(const float* src, int C, int T, int B, float* dst_f32, void* dst_op, int prec, int ldd, int cpad, hipStream_t s,
                             int ldd_op = 0, int split = 0);   // split > 0 (16-bit): dst_op rows of ldd_op columns hold a hi + lo pair, lo at column split + c
hipError_t launch_btc_to_nct(const float* src, int lds, int C, int T, int B, float* dst, hipStream_t s);
hipError_t launch_mask_bias(const uint8_t* mask, int n, float* bias, hipStream_t s);
hipError_t launch_ln_apply(const float* x, int M, int C, float eps, const float* gamma, const float* beta,
                           float* out, int L, int Lout_stride_rows, hipStream_t s);
hipError_t launch_pool_cls(float* seq, int B, int L, int C, const float* pos, hipStream_t s);
hipError_t launch_pool_attn(const float* qkv, int B, int L1, int C, int heads, float* pooled, hipStream_t s);
hipError_t launch_pool_proj(const float* pooled, int B, int C, const float* wt, const float* b, int E,
                            const float* gamma, const float* beta, float eps, float* out, hipStream_t s);
hipError_t launch_solver_update(const float* coef, const int* step_ptr, int ncoef, const float* x0,
                                float* xe, void* xe_op, int prec, float* xbar, float* d1, float* mprev, size_t n, hipStream_t s,
                                int split = 0);                 // split > 0 (16-bit): xe_op rows are [hi(split) | lo(split)] of state rows of `split` columns
hipError_t launch_fill_i32(int* p, int v, hipStream_t s);
hipError_t launch_placement(unsigned* dev_out, int n_blocks, int spin, hipStream_t s);
hipError_t launch_snapshot_u32(unsigned* src, unsigned* dst, hipStream_t s);   // *dst = atomicExch(src, 0)

}  // namespace ns2vc
