// Flash-style multi-head attention for the NS2VC denoiser on CDNA4 (gfx950).
//
// Replaces F.scaled_dot_product_attention at reference
// unet1d/attention_processor.py:1032 (self-attention, Lq = Lk = T_l, no mask; and
// prompt cross-attention, Lk = Lp, additive mask bias (1-m)*-10000 built at
// unet1d/unet_1d_condition.py:816-818 and broadcast over heads :1003-1007).
//
// One workgroup = 4 waves = 128 queries of one (batch, head); each wave owns 32
// queries.  K/V stream through LDS in 64-key tiles (register prefetch of tile
// t+1 under the MFMAs of tile t, one barrier per tile).  The score tile is
// computed TRANSPOSED, S^T = K * Q^T, so that after the 32x32 MFMA each lane
// holds 16 keys of ONE query (col = lane&31): the online-softmax row reduction is
// lane-local plus one exchange with lane^32, and the probabilities feed the PV
// MFMA's B operand straight from registers.  O^T = V^T * P^T needs V^T tiles,
// which are written transposed into LDS at staging time.
//   bf16: v_mfma_f32_32x32x16_bf16, P rounded to bf16, fp32 softmax state / accumulators
//   f32 : v_mfma_f32_32x32x2_f32 (exact fp32, parity mode)
// head_dim in {16,32,48,64} (NS2VC: C_l/8 for C_l in {128,256,384,512}).
#include "common.h"

namespace ns2vc {

template <typename T> struct AMma;
template <> struct AMma<float> {
  static constexpr int SZ = 4;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
  // position (in elements) of key `key` (0..31) inside a 32-key V^T sub-row
  __device__ static __forceinline__ int vpos(int key) { return key; }
};
template <> struct AMma<bf16_t> {
  static constexpr int SZ = 2;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const uint4& a, const uint4& b) {
    union U { uint4 u; bf16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
  // swap key bits 2 and 3 so that the 8 keys one lane-half contributes to a
  // 16-key MFMA k-slab are contiguous (see header comment of attn_kernel)
  __device__ static __forceinline__ int vpos(int key) { return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1); }
};

template <typename TM> __device__ __forceinline__ void lds_store4(char* p, const float4& v);
template <> __device__ __forceinline__ void lds_store4<float>(char* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void lds_store4<bf16_t>(char* p, const float4& v) {
  uint2 r;
  r.x = (uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16);
  r.y = (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16);
  *reinterpret_cast<uint2*>(p) = r;
}
template <typename TM> __device__ __forceinline__ void lds_store1(char* p, float v);
template <> __device__ __forceinline__ void lds_store1<float>(char* p, float v) { *reinterpret_cast<float*>(p) = v; }
template <> __device__ __forceinline__ void lds_store1<bf16_t>(char* p, float v) { *reinterpret_cast<uint16_t*>(p) = f32_to_bf16_bits(v); }

template <typename TM, int HD>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
  constexpr int SZ = AMma<TM>::SZ;
  constexpr int EPC = 16 / SZ;            // elements per 16-B fragment chunk
  constexpr int NS = HD * SZ / 32;        // 32-B d-slabs per key row (QK^T k-steps)
  constexpr int KROWB = HD * SZ + 16;     // K tile row bytes   (stride = 4*odd dwords)
  constexpr int VROWB = 64 * SZ + 16;     // V^T tile row bytes (64 keys)
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int DT = HDP / 32;
  constexpr int NSL = SZ;                 // 32-B key-slabs per 32-key sub-tile (f32: 4x8 keys, bf16: 2x16 keys)
  constexpr int KBYTES = 64 * KROWB, VBYTES = HDP * VROWB;
  constexpr int STAGE = KBYTES + VBYTES + 64 * 4;
  constexpr int UPT = HD / 16;            // float4 units per thread per tile (K and V each)
  constexpr int QPR = HD / 4;             // float4 units per key row
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const float LOG2E = 1.4426950408889634f;

  // zero both stages once (V^T pad rows d >= HD must read as 0)
  for (int i = tid * 16; i < 2 * STAGE; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q, hi) holds d = s*2*EPC + hi*EPC .. +EPC
  uint4 qf[NS];
  {
    const float sc = a.scale * LOG2E;
    const float* qp = reinterpret_cast<const float*>(a.q) + ((size_t)(b * a.Lq + min(q, a.Lq - 1)) * a.ldq + h * HD);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        float4 t = *reinterpret_cast<const float4*>(qp + s * 2 * EPC + hi * EPC + e);
        if (q >= a.Lq) t = make_float4(0.f, 0.f, 0.f, 0.f);
        v[e] = t.x * sc; v[e + 1] = t.y * sc; v[e + 2] = t.z * sc; v[e + 3] = t.w * sc;
      }
      if constexpr (SZ == 4) {
        qf[s] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
      } else {
        qf[s].x = (uint32_t)f32_to_bf16_bits(v[0]) | ((uint32_t)f32_to_bf16_bits(v[1]) << 16);
        qf[s].y = (uint32_t)f32_to_bf16_bits(v[2]) | ((uint32_t)f32_to_bf16_bits(v[3]) << 16);
        qf[s].z = (uint32_t)f32_to_bf16_bits(v[4]) | ((uint32_t)f32_to_bf16_bits(v[5]) << 16);
        qf[s].w = (uint32_t)f32_to_bf16_bits(v[6]) | ((uint32_t)f32_to_bf16_bits(v[7]) << 16);
      }
    }
  }

  const float* kbase = reinterpret_cast<const float*>(a.k) + (size_t)b * a.Lk * a.ldk + h * HD;
  const float* vbase = reinterpret_cast<const float*>(a.v) + (size_t)b * a.Lk * a.ldv + h * HD;
  const float* bias = a.bias ? a.bias + (size_t)b * a.Lk : nullptr;

  float4 kraw[UPT], vraw[UPT];
  float braw = 0.f;
  auto load_tile = [&](int t) {
    const int key0 = t * 64;
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + 256 * i;
      {  // K: row-major units, coalesced along d
        const int key = u / QPR, dq = u - key * QPR;
        const int kk = key0 + key;
        kraw[i] = (kk < a.Lk) ? *reinterpret_cast<const float4*>(kbase + (size_t)kk * a.ldk + dq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      {  // V: key-fastest units (transposed LDS write is then conflict-free)
        const int key = u & 63, dq = u >> 6;
        const int kk = key0 + key;
        vraw[i] = (kk < a.Lk) ? *reinterpret_cast<const float4*>(vbase + (size_t)kk * a.ldv + dq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (tid < 64) {
      const int kk = key0 + tid;
      braw = (kk < a.Lk) ? (bias ? bias[kk] * LOG2E : 0.f) : -INFINITY;
    }
  };
  auto store_tile = [&](int stage) {
    char* Ks = smem + stage * STAGE;
    char* Vs = Ks + KBYTES;
    float* Bs = reinterpret_cast<float*>(Vs + VBYTES);
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + 256 * i;
      {
        const int key = u / QPR, dq = u - key * QPR;
        lds_store4<TM>(Ks + key * KROWB + dq * 4 * SZ, kraw[i]);
      }
      {
        const int key = u & 63, dq = u >> 6;
        const int pos = (key & 32) + AMma<TM>::vpos(key & 31);
        char* vp = Vs + (dq * 4) * VROWB + pos * SZ;
        lds_store1<TM>(vp, vraw[i].x);
        lds_store1<TM>(vp + VROWB, vraw[i].y);
        lds_store1<TM>(vp + 2 * VROWB, vraw[i].z);
        lds_store1<TM>(vp + 3 * VROWB, vraw[i].w);
      }
    }
    if (tid < 64) Bs[tid] = braw;
  };

  f32x16_t o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int ntile = (a.Lk + 63) / 64;
  __syncthreads();          // zero-fill done
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) load_tile(t + 1);
    const char* Ks = smem + (t & 1) * STAGE;
    const char* Vs = Ks + KBYTES;
    const float* Bs = reinterpret_cast<const float*>(Vs + VBYTES);

    // ---- S^T[key][q] = sum_d K[key][d] * Q[q][d]   (two 32-key sub-tiles)
    f32x16_t s[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[k2][r] = 0.f;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const uint4 kf = *reinterpret_cast<const uint4*>(Ks + (k2 * 32 + l31) * KROWB + sl * 32 + hi * 16);
        AMma<TM>::mma(s[k2], kf, qf[sl]);
      }
    }
    // ---- additive bias (mask) and tail-key masking; lane's keys: k2*32 + 8*g + 4*hi + i
    const bool need_bias = (bias != nullptr) || (t * 64 + 64 > a.Lk);
    if (need_bias) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 bb = *reinterpret_cast<const float4*>(Bs + k2 * 32 + 8 * gq + 4 * hi);
          s[k2][4 * gq + 0] += bb.x; s[k2][4 * gq + 1] += bb.y; s[k2][4 * gq + 2] += bb.z; s[k2][4 * gq + 3] += bb.w;
        }
    }
    // ---- online softmax (base-2), state per query = per lane (both lane halves agree on m)
    float mx = s[0][0];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[k2][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = exp2f(s[k2][r] - m_new);
        s[k2][r] = p;
        psum += p;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // ---- O^T[d][q] += sum_key V^T[d][key] * P^T[key][q]
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        uint4 pf;
        if constexpr (SZ == 4) {
          pf = make_uint4(__float_as_uint(s[k2][4 * sl + 0]), __float_as_uint(s[k2][4 * sl + 1]),
                          __float_as_uint(s[k2][4 * sl + 2]), __float_as_uint(s[k2][4 * sl + 3]));
        } else {
          pf.x = (uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 0]) | ((uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 1]) << 16);
          pf.y = (uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 2]) | ((uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 3]) << 16);
          pf.z = (uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 4]) | ((uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 5]) << 16);
          pf.w = (uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 6]) | ((uint32_t)f32_to_bf16_bits(s[k2][8 * sl + 7]) << 16);
        }
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const uint4 vf = *reinterpret_cast<const uint4*>(Vs + (d * 32 + l31) * VROWB + k2 * 32 * SZ + sl * 32 + hi * 16);
          AMma<TM>::mma(o[d], vf, pf);
        }
      }
    }
    if (t + 1 < ntile) store_tile((t + 1) & 1);
    __syncthreads();
  }

  // ---- normalise and store O[q][h*HD + d]; lane holds d = dt*32 + 8*g + 4*hi + i
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q < a.Lq) {
    float* op = reinterpret_cast<float*>(a.out) + ((size_t)(b * a.Lq + q) * a.ldo + h * HD);
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d0 = d * 32 + 8 * gq + 4 * hi;
        if (d0 < HD) {
          *reinterpret_cast<float4*>(op + d0) =
              make_float4(o[d][4 * gq] * inv, o[d][4 * gq + 1] * inv, o[d][4 * gq + 2] * inv, o[d][4 * gq + 3] * inv);
        }
      }
  }
}

template <typename TM, int HD> static constexpr size_t attn_lds() {
  constexpr int SZ = AMma<TM>::SZ;
  constexpr int HDP = (HD + 31) / 32 * 32;
  return 2 * (size_t)(64 * (HD * SZ + 16) + HDP * (64 * SZ + 16) + 64 * 4);
}

template <typename TM, int HD> static hipError_t launch_hd(const AttnArgs& a, hipStream_t s) {
  dim3 grid((a.Lq + 127) / 128, a.H, a.B);
  const size_t lds = attn_lds<TM, HD>();
  hipLaunchKernelGGL((attn_kernel<TM, HD>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}

template <typename TM> static hipError_t launch_tm(const AttnArgs& a, int hd, hipStream_t s) {
  switch (hd) {
    case 16: return launch_hd<TM, 16>(a, s);
    case 32: return launch_hd<TM, 32>(a, s);
    case 48: return launch_hd<TM, 48>(a, s);
    case 64: return launch_hd<TM, 64>(a, s);
    default: return hipErrorInvalidValue;
  }
}

template <typename TM, int HD> static hipError_t set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<TM, HD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)attn_lds<TM, HD>());
}
hipError_t init_attn_attributes() {
  hipError_t e;
  if ((e = set_attr<float, 16>()) != hipSuccess) return e;
  if ((e = set_attr<float, 32>()) != hipSuccess) return e;
  if ((e = set_attr<float, 48>()) != hipSuccess) return e;
  if ((e = set_attr<float, 64>()) != hipSuccess) return e;
  if ((e = set_attr<bf16_t, 16>()) != hipSuccess) return e;
  if ((e = set_attr<bf16_t, 32>()) != hipSuccess) return e;
  if ((e = set_attr<bf16_t, 48>()) != hipSuccess) return e;
  if ((e = set_attr<bf16_t, 64>()) != hipSuccess) return e;
  return hipSuccess;
}

hipError_t launch_attention(const AttnArgs& a, int head_dim, int prec, hipStream_t s) {
  if (a.Lq <= 0 || a.Lk <= 0 || (a.ldq & 3) || (a.ldk & 3) || (a.ldv & 3) || (a.ldo & 3)) return hipErrorInvalidValue;
  return prec == PREC_BF16 ? launch_tm<bf16_t>(a, head_dim, s) : launch_tm<float>(a, head_dim, s);
}

}  // namespace ns2vc
