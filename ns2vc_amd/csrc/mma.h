// MFMA / LDS-DMA helpers of the GEMM kernels (gfx950).
#pragma once
#include "common.h"

namespace ns2vc {



template <typename T> struct MmaT;
template <> struct MmaT<float> {
  static constexpr int EPC = 4;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct MmaT<bf16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    union U { u32x4_t u; bf16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
};

template <> struct MmaT<f16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    union U { u32x4_t u; f16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.v, ub.v, acc, 0, 0, 0);
  }
};

// direct HBM/L2 -> LDS DMA of 16 B per lane: LDS address = lds_dst (wave-uniform) + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// The same DMA through a buffer descriptor: address = rsrc.base + voff (per lane) + soff (wave-uniform SGPR).  Lanes whose
// voff is out of the descriptor's range deliver ZEROS to LDS (checked on MI355X by tools/dma_probe.hip), which is how
// padded taps / rows past M are produced, and the per-tile K offset rides in the SGPR: no per-lane address arithmetic at all.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned DMA_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ i32x4_t make_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  i32x4_t r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));     // stride 0: raw buffer
  r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)(bytes > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : bytes));
  r.w = 0x00020000;
  return r;
}
__device__ __forceinline__ i32x4_t uniform_rsrc(const i32x4_t r) {   // pin a wave-uniform descriptor into SGPRs
  i32x4_t u;
  u.x = __builtin_amdgcn_readfirstlane(r.x); u.y = __builtin_amdgcn_readfirstlane(r.y);
  u.z = __builtin_amdgcn_readfirstlane(r.z); u.w = __builtin_amdgcn_readfirstlane(r.w);
  return u;
}
__device__ __forceinline__ void blds16(const i32x4_t rsrc_, unsigned voff, unsigned soff_, unsigned lds_dst_) {
  // (readfirstlane of an already-uniform value folds away; it only tells the register allocator "SGPR")
  const i32x4_t rsrc = uniform_rsrc(rsrc_);
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)soff_);
  const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }


}  // namespace ns2vc
