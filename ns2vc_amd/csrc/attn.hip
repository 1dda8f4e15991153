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
// q/k/v/out are operand-typed tensors (bf16, or fp32 in parity mode): no conversions while staging.
#include "common.h"
#include "mma.h"
#include <cstdlib>

namespace ns2vc {

template <typename T> struct AMma;
template <> struct AMma<float> {
  static constexpr int SZ = 4;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x16_t mma0(const u32x4_t& a, const u32x4_t& b) {   // acc = a*b (srcC = inline 0)
    f32x16_t acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    return acc;
  }
  // position (in elements) of key `key` (0..31) inside a 32-key V^T sub-row
  __device__ static __forceinline__ int vpos(int key) { return key; }
};
template <typename TM> struct AMma16 {     // bf16_t / f16_t: one v_mfma_f32_32x32x16_{bf16,f16} per 32-B k-slab
  static constexpr int SZ = 2;
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) { MmaT<TM>::mma(acc, a, b); }
  __device__ static __forceinline__ f32x16_t mma0(const u32x4_t& a, const u32x4_t& b) {   // acc = a*b (srcC = inline 0)
    f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    MmaT<TM>::mma(acc, a, b);
    return acc;
  }
  // swap key bits 2 and 3 so that the 8 keys one lane-half contributes to a
  // 16-key MFMA k-slab are contiguous (see header comment of attn_kernel)
  __device__ static __forceinline__ int vpos(int key) { return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1); }
};
template <> struct AMma<bf16_t> : AMma16<bf16_t> {};
template <> struct AMma<f16_t> : AMma16<f16_t> {};

// scatter the EPC elements of one 16-B piece (one key, EPC consecutive d) down a V^T column
template <typename TM> __device__ __forceinline__ void vt_scatter(char* vp, int rowb, const u32x4_t& v);
template <> __device__ __forceinline__ void vt_scatter<float>(char* vp, int rowb, const u32x4_t& v) {
  *reinterpret_cast<uint32_t*>(vp) = v.x;
  *reinterpret_cast<uint32_t*>(vp + rowb) = v.y;
  *reinterpret_cast<uint32_t*>(vp + 2 * rowb) = v.z;
  *reinterpret_cast<uint32_t*>(vp + 3 * rowb) = v.w;
}
__device__ __forceinline__ void vt_scatter16(char* vp, int rowb, const u32x4_t& v) {
  *reinterpret_cast<uint16_t*>(vp) = (uint16_t)v.x;
  *reinterpret_cast<uint16_t*>(vp + rowb) = (uint16_t)(v.x >> 16);
  *reinterpret_cast<uint16_t*>(vp + 2 * rowb) = (uint16_t)v.y;
  *reinterpret_cast<uint16_t*>(vp + 3 * rowb) = (uint16_t)(v.y >> 16);
  *reinterpret_cast<uint16_t*>(vp + 4 * rowb) = (uint16_t)v.z;
  *reinterpret_cast<uint16_t*>(vp + 5 * rowb) = (uint16_t)(v.z >> 16);
  *reinterpret_cast<uint16_t*>(vp + 6 * rowb) = (uint16_t)v.w;
  *reinterpret_cast<uint16_t*>(vp + 7 * rowb) = (uint16_t)(v.w >> 16);
}
template <> __device__ __forceinline__ void vt_scatter<bf16_t>(char* vp, int rowb, const u32x4_t& v) { vt_scatter16(vp, rowb, v); }
template <> __device__ __forceinline__ void vt_scatter<f16_t>(char* vp, int rowb, const u32x4_t& v) { vt_scatter16(vp, rowb, v); }

// combine a value with its partner lane (lane ^ 32) without an LDS round trip: v_permlane32_swap puts the lower
// half's values in one result and the upper half's in the other, for every lane
__device__ __forceinline__ float half_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// operand-type helpers of the softmax reference shift
template <typename TM> __device__ __forceinline__ float round_op(float x);
template <> __device__ __forceinline__ float round_op<float>(float x) { return x; }
template <> __device__ __forceinline__ float round_op<bf16_t>(float x) { return __uint_as_float(pack_bf16x2(0.f, x) & 0xffff0000u); }
template <> __device__ __forceinline__ float round_op<f16_t>(float x) { return f16_lo(pack_f16x2(x, 0.f)); }
// first 16-B chunk of a K/Q "aux" k-slab: {e0, e1, 0, ...}
template <typename TM> __device__ __forceinline__ u32x4_t aux_chunk(float e0, float e1);
template <> __device__ __forceinline__ u32x4_t aux_chunk<float>(float e0, float e1) { return u32x4_t{__float_as_uint(e0), __float_as_uint(e1), 0u, 0u}; }
template <> __device__ __forceinline__ u32x4_t aux_chunk<bf16_t>(float e0, float e1) { return u32x4_t{pack_bf16x2(e0, e1), 0u, 0u, 0u}; }
template <> __device__ __forceinline__ u32x4_t aux_chunk<f16_t>(float e0, float e1) { return u32x4_t{pack_f16x2(e0, e1), 0u, 0u, 0u}; }
template <typename TM> __device__ __forceinline__ TM op_from_float(float x);
template <> __device__ __forceinline__ float op_from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t op_from_float<bf16_t>(float x) { bf16_t r; r.v = (uint16_t)pack_bf16x2(x, 0.f); return r; }
template <> __device__ __forceinline__ f16_t op_from_float<f16_t>(float x) { f16_t r; r.v = (uint16_t)pack_f16x2(x, 0.f); return r; }   // (-inf passes)

// The attention kernel is VALU-bound on MI355X (v_exp_f32 is quarter rate and every score used to cost a scale-fma,
// a max, a subtract, an exp, a sum-add and half a convert), while its MFMA pipe idles.  So everything except max / exp /
// convert is pushed INTO the MFMAs:
//   * the softmax scale (in log2 units) is folded into Q once, at load;
//   * every K row carries an extra "aux" k-slab {1, bias(key), 0...} and every Q fragment the matching {-m_ref, 1, 0...}:
//     the score MFMA returns  s*scale*log2e + bias - m_ref  directly (mask bias and tail-key -inf included);
//   * m_ref is a per-query REFERENCE, not the running maximum: probabilities are exp2(score - m_ref) for a whole run of
//     tiles and only when some score exceeds m_ref by 2^12 are accumulators and reference moved (softmax is
//     shift-invariant, so any common reference is exact; it is kept representable in the operand type so the shift done
//     by the MFMA is exact too);
//   * V^T carries a row of ones, so the PV MFMA also accumulates the denominator (of the SAME rounded probabilities
//     that build the numerator): no per-score add, no separate running sum.
// P8 (16-bit operand types only; BASELINE config 5's "fp8 MFMA attention path"): the PV product runs on the fp8-RATE matrix
// instruction of gfx950, v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 = one whole key tile per instruction, OCP e4m3 operands, unit
// E8M0 scales; twice the bf16 rate -- the non-scaled v_mfma_f32_32x32x16_fp8_fp8 of round 2 runs at the bf16 rate,
// MI355X_MICROARCH.md "Matrix cores").  V^T is staged in LDS as fp8 (converted while staging), the probabilities are packed to
// fp8 instead of the 16-bit type; QK^T (K = head dim: 16..64), the softmax state and all accumulators are unchanged.
// Operand slots: the instruction contracts over 64 k-slots, lane half `hi` supplying slots 32*hi .. 32*hi+31 of its A row / B
// column (8 VGPRs, 4 slots each).  The contraction index is a dummy, so the keys are PERMUTED to where the score MFMA left
// them: after S^T = K Q^T lane (q, hi) holds the keys  k2*32 + 8*g + 4*hi + i  (k2 = 32-key sub-tile, g = 0..3, i = 0..3) in
// s[k2][4*g + i]; key -> slot 32*hi + (16*k2 + 4*g + i), i.e. P's register v = 4*k2 + g is pack(s[k2][4g .. 4g+3]) with no
// data movement at all, and V^T rows are stored with the same key -> slot map (vpos8).  See DESIGN section 4 for why it stays an
// option (the kernel is VALU-bound, fp8 shortens only the matrix time; 3 mantissa bits in P cost the parity bar).
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  return (uint32_t)v;
}
typedef int i32x8_t __attribute__((ext_vector_type(8)));
#ifndef NS2VC_ATTN_OPT
#define NS2VC_ATTN_OPT 1
#endif
template <bool B> struct OptTag { static constexpr bool value = B; };
// test hook (ns2vc_debug_set_attn_optimistic): 1 = every workgroup takes the exact pass only
__device__ int g_attn_exact_only = 0;
// slot of key (0..63) inside a 64-slot V^T row of the fp8 path
__device__ __forceinline__ int vpos8(int key) { return ((key & 4) << 3) | ((key & 32) >> 1) | ((key & 24) >> 1) | (key & 3); }
template <typename TM, int HD, int KEYS, bool P8>
__global__ __launch_bounds__(256) void attn_kernel(const AttnArgs a) {
  op_mode_init<TM>();
  static_assert(!P8 || (sizeof(TM) == 2 && KEYS == 64), "the fp8 PV path: 16-bit operand types, one 64-key tile per f8f6f4 MFMA");
  constexpr int NSUB = KEYS / 32;         // 32-key sub-tiles per K/V tile (2 or 4): per-tile bookkeeping, barrier and waits amortise over them
  constexpr int SZ = AMma<TM>::SZ;
  constexpr int EPC = 16 / SZ;            // elements per 16-B fragment chunk
  constexpr int NS = HD * SZ / 32;        // 32-B d-slabs per key row (QK^T k-steps), + 1 aux slab
  constexpr int KROWB = HD * SZ + 48;     // K tile row bytes: HD elements, 32-B aux slab, pad (stride = 4*odd dwords)
  constexpr int VSZ = P8 ? 1 : SZ;        // bytes per V^T / P element
  constexpr int VROWB = KEYS * VSZ + 16;  // V^T tile row bytes (KEYS keys)
  constexpr int HDX = (HD + 1 + 31) / 32 * 32;   // V^T rows: HD value rows + the ones row (row HD), padded to 32
  constexpr int DT = HDX / 32;
  constexpr int NSL = SZ;                 // 32-B key-slabs per 32-key sub-tile (f32: 4x8 keys, bf16: 2x16 keys)
  constexpr int KBYTES = KEYS * KROWB, VBYTES = HDX * VROWB;
  constexpr int STAGE = KBYTES + VBYTES;
  constexpr int PPR = HD * SZ / 16;       // 16-B pieces per key row
  constexpr int NPIECE = KEYS * PPR;      // pieces per K (or V) tile
  constexpr int UPT = (NPIECE + 255) / 256;
  constexpr float THRESH = P8 ? 8.0f : 12.0f;   // move the reference when a score exceeds it by more than this (log2 units; fp8: p <= 2^8 < 448)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // XCD-aware mapping: consecutive workgroup ids round-robin over the 8 XCDs (each with a private L2), so the ids are
  // remapped to give every XCD a contiguous run of (batch item, head, query block): all heads and query blocks of one
  // batch item share K/V rows (a head is a 32..128-B column slice of them) and now hit the same L2.  Measured before
  // the remap (rocprofv3 FETCH_SIZE): 77-132 MB fetched per launch for ~28 MB of Q/K/V, L2 hit rate 24-29 %.
  const int nqb = (a.Lq + 127) >> 7;
  int b, h, qb;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int qn = nwg >> 3, rn = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    b = swz / (a.H * nqb);
    const int rem = swz - b * (a.H * nqb);
    h = rem / nqb;
    qb = rem - h * nqb;
  }
  const int q = qb * 128 + wave * 32 + l31;
  const float LOG2E = 1.4426950408889634f;
  const float sc2 = a.scale * LOG2E;      // scores are kept in log2 units

  // zero both stages once (V^T pad rows and the aux slabs' tails must read as 0), then the constants
  for (int i = tid * 16; i < 2 * STAGE; i += 256 * 16) *reinterpret_cast<u32x4_t*>(smem + i) = u32x4_t{0, 0, 0, 0};
  __syncthreads();
  for (int i = tid; i < 2 * KEYS; i += 256) {      // K aux element 0 = 1 for every key row, V^T row HD = ones, both stages
    const int st = i / KEYS, key = i - st * KEYS;
    *reinterpret_cast<TM*>(smem + st * STAGE + key * KROWB + HD * SZ) = op_from_float<TM>(1.0f);
    if constexpr (P8) *reinterpret_cast<uint8_t*>(smem + st * STAGE + KBYTES + HD * VROWB + key) = (uint8_t)0x38;     // e4m3 1.0; all 64 slots of the row
    else *reinterpret_cast<TM*>(smem + st * STAGE + KBYTES + HD * VROWB + key * SZ) = op_from_float<TM>(1.0f);
  }

  // ---- Q fragments (B operand of S^T = K Q^T), pre-multiplied by scale*log2e: lane (q, hi) holds d = s*2*EPC + hi*EPC .. +EPC
  u32x4_t qf[NS];
  {
    const TM* qp = reinterpret_cast<const TM*>(a.q) + ((size_t)(b * a.Lq + min(q, a.Lq - 1)) * a.ldq + h * HD);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qp + s * 2 * EPC + hi * EPC);
      if constexpr (SZ == 4) {
        qf[s] = u32x4_t{__float_as_uint(__uint_as_float(raw.x) * sc2), __float_as_uint(__uint_as_float(raw.y) * sc2),
                        __float_as_uint(__uint_as_float(raw.z) * sc2), __float_as_uint(__uint_as_float(raw.w) * sc2)};
      } else {
        auto sc = [&](uint32_t w) __attribute__((always_inline)) { return Op16<TM>::pack(Op16<TM>::lo(w) * sc2, Op16<TM>::hi(w) * sc2); };
        qf[s] = u32x4_t{sc(raw.x), sc(raw.y), sc(raw.z), sc(raw.w)};
      }
      if (q >= a.Lq) qf[s] = u32x4_t{0, 0, 0, 0};
    }
  }
  float m_ref = 0.f;                       // softmax reference of this lane's query (operand-representable)
  u32x4_t qaux = hi == 0 ? aux_chunk<TM>(-m_ref, 1.0f) : u32x4_t{0, 0, 0, 0};
  // r6, hd 16 without a mask bias (the level-0 self-attention: a quarter of this kernel's MFMAs is the aux slab, which then only carries -m_ref): the
  // reference enters as the score MFMA's C operand instead -- sixteen registers that hold -m_ref, rewritten when the reference moves (once per pass in the
  // optimistic form) -- and the aux slab, its LDS read and its MFMA are left to the one tile that needs the tail keys' -inf.  Same sum in the same order:
  // the aux product was the first addend of every score already.  (hd 32: the sixteen registers would cost a wave per SIMD; with a bias the C operand
  // would have to be rebuilt per tile, 16 VALU adds against one MFMA.)
  constexpr bool CINIT = HD == 16 && SZ == 2 && !P8;
  f32x16_t cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = 0.f;

  const TM* kbase = reinterpret_cast<const TM*>(a.k) + (size_t)b * a.Lk * a.ldk + h * HD;
  const TM* vbase = reinterpret_cast<const TM*>(a.v) + (size_t)b * a.Lk * a.ldv + h * HD;
  const float* bias = a.bias ? a.bias + (size_t)b * a.Lk : nullptr;

  u32x4_t kraw[UPT], vraw[UPT];
  float braw = 0.f;
  // Rows past Lk are fetched from the LAST valid row instead of being replaced by zeros: their scores carry the -inf of the
  // aux slab whatever K holds (finite), and their probabilities are exactly 0 against any finite V -- so the loads need no
  // per-lane select / zero-fill and no divergent branch.  The mask bias is kept RAW here and scaled in store_tile: scaling it
  // at load time made the compiler wait for this tile's whole prefetch (s_waitcnt vmcnt(0)) in front of the MFMAs.
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int key0 = t * KEYS;
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + 256 * i;
      if (u < NPIECE) {
        {  // K: row-major pieces, coalesced along d
          const int key = u / PPR, pc = u - key * PPR;
          const int kk = min(key0 + key, a.Lk - 1);
          kraw[i] = *reinterpret_cast<const u32x4_t*>(kbase + (size_t)kk * a.ldk + pc * EPC);
        }
        {  // V: key-fastest pieces (the transposed LDS write is then conflict-free)
          const int key = u % KEYS, pc = u / KEYS;
          const int kk = min(key0 + key, a.Lk - 1);
          vraw[i] = *reinterpret_cast<const u32x4_t*>(vbase + (size_t)kk * a.ldv + pc * EPC);
        }
      }
    }
    if (tid < KEYS && bias) braw = bias[min(key0 + tid, a.Lk - 1)];
  };
  auto store_tile = [&](int stage, int key0s) __attribute__((always_inline)) {
    char* Ks = smem + stage * STAGE;
    char* Vs = Ks + KBYTES;
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      const int u = tid + 256 * i;
      if (u < NPIECE) {
        {
          const int key = u / PPR, pc = u - key * PPR;
          *reinterpret_cast<u32x4_t*>(Ks + key * KROWB + pc * 16) = kraw[i];
        }
        {
          const int key = u % KEYS, pc = u / KEYS;
          if constexpr (P8) {       // eight 16-bit values of one key -> eight fp8 bytes down the V^T column, at the key's k-slot
            const u32x4_t w = vraw[i];
            const uint32_t b0 = pack_fp8x4(Op16<TM>::lo(w.x), Op16<TM>::hi(w.x), Op16<TM>::lo(w.y), Op16<TM>::hi(w.y));
            const uint32_t b1 = pack_fp8x4(Op16<TM>::lo(w.z), Op16<TM>::hi(w.z), Op16<TM>::lo(w.w), Op16<TM>::hi(w.w));
            uint8_t* vp = reinterpret_cast<uint8_t*>(Vs + (pc * EPC) * VROWB + vpos8(key));
#pragma unroll
            for (int j = 0; j < 4; ++j) { vp[j * VROWB] = (uint8_t)(b0 >> (8 * j)); vp[(4 + j) * VROWB] = (uint8_t)(b1 >> (8 * j)); }
          } else {
            const int pos = (key & ~31) + AMma<TM>::vpos(key & 31);
            vt_scatter<TM>(Vs + (pc * EPC) * VROWB + pos * SZ, VROWB, vraw[i]);
          }
        }
      }
    }
    if (tid < KEYS) {                                 // K aux element 1 = bias(key) in log2 units; -inf for the keys past Lk
      const float bl = (key0s + tid < a.Lk) ? (bias ? braw * LOG2E : 0.f) : -INFINITY;
      *reinterpret_cast<TM*>(Ks + tid * KROWB + HD * SZ + SZ) = op_from_float<TM>(bl);
    }
  };

  f32x16_t o[DT];
  const int ntile = (a.Lk + KEYS - 1) / KEYS;
  // OPT (r3, 16-bit operand types without the fp8 PV): the per-tile maximum (16 v_max3 + a lane exchange + a vote per tile, ~15 %
  // of the loop's VALU work, which is what bounds this kernel) only guards the 16-bit range of the probabilities.  The optimistic
  // pass takes the reference from the FIRST tile alone (its maximum + OPT_MARGIN: later scores may exceed the first tile's by
  // 2^(14 + margin) before the check at the end sends the row to the exact pass) and checks nothing afterwards; an overflow (or a
  // row whose probabilities all flushed to zero behind a fully masked first tile) shows up in the denominator, which the PV MFMA
  // accumulates anyway, and sends the whole workgroup through the exact pass below.  Softmax is shift-invariant: both passes are
  // exact up to the rounding of the probabilities.
  constexpr bool OPT = NS2VC_ATTN_OPT && !P8 && sizeof(TM) == 2;
  constexpr float OPT_MARGIN = 4.0f;
  // `compute` (wave-uniform): false = this wave keeps the accumulators it has and only helps staging the K / V tiles (r4: the exact pass of
  // a workgroup is computed by the waves that hold a failing row; the others would only burn the VALU slots that bound this kernel)
  auto run = [&](auto optimistic, const bool compute) __attribute__((always_inline)) {
  constexpr bool O = decltype(optimistic)::value;
  if (compute) {
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    m_ref = 0.f;
    qaux = hi == 0 ? aux_chunk<TM>(-m_ref, 1.0f) : u32x4_t{0, 0, 0, 0};
    if constexpr (CINIT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    }
  }
  load_tile(0);
  __syncthreads();          // constants written (first pass) / everybody is done with the stages (second pass)
  store_tile(0, 0);
  __syncthreads();

  for (int t = 0; t < ntile; ++t) {
    if (t + 1 < ntile) load_tile(t + 1);
    const char* Ks = smem + (t & 1) * STAGE;
    const char* Vs = Ks + KBYTES;
    if (compute) {

    // ---- S'^T[key][q] = sum_d K[key][d] * (Q[q][d]*scale*log2e) + 1*(-m_ref[q]) + bias[key]*1   (two 32-key sub-tiles)
    f32x16_t s[NSUB];
    if (CINIT && !bias && (t + 1 < ntile || a.Lk % KEYS == 0)) {       // (wave-uniform) the reference through the C operand, no aux slab
#pragma unroll
      for (int k2 = 0; k2 < NSUB; ++k2) {
        const char* kr = Ks + (k2 * 32 + l31) * KROWB + hi * 16;
        s[k2] = cinit;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) AMma<TM>::mma(s[k2], *reinterpret_cast<const u32x4_t*>(kr + sl * 32), qf[sl]);
      }
    } else {
#pragma unroll
    for (int k2 = 0; k2 < NSUB; ++k2) {
      const char* kr = Ks + (k2 * 32 + l31) * KROWB + hi * 16;
      s[k2] = AMma<TM>::mma0(*reinterpret_cast<const u32x4_t*>(kr + NS * 32), qaux);      // aux slab first: no zero-init movs
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) AMma<TM>::mma(s[k2], *reinterpret_cast<const u32x4_t*>(kr + sl * 32), qf[sl]);
    }
    }
    // ---- reference check (per query = per lane; both lane halves agree): lane's keys are k2*32 + 8*g + 4*hi + i
    if (!O || t == 0) {
      float mx = s[0][0];
#pragma unroll
      for (int k2 = 0; k2 < NSUB; ++k2)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[k2][r]);
      mx = half_max(mx);
      if (t == 0 || __any(mx > THRESH)) {                 // wave-uniform; after the first tile this is rare
        float target = t == 0 ? mx : fmaxf(mx, 0.f);
        if (!(target > -INFINITY)) target = 0.f;          // nothing but masked keys so far
        if (O) target += OPT_MARGIN;
        const float m_new = round_op<TM>(m_ref + target);
        const float dsh = m_new - m_ref;
        if (t > 0) {
          const float alpha = __builtin_amdgcn_exp2f(-dsh);    // dsh >= 0
#pragma unroll
          for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
#pragma unroll
        for (int k2 = 0; k2 < NSUB; ++k2)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[k2][r] -= dsh;
        m_ref = m_new;
        if (hi == 0) qaux = aux_chunk<TM>(-m_ref, 1.0f);
        if constexpr (CINIT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) cinit[r] = -m_ref;
        }
      }
    }
#pragma unroll
    for (int k2 = 0; k2 < NSUB; ++k2)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[k2][r] = __builtin_amdgcn_exp2f(s[k2][r]);

    // ---- [O^T ; l][d][q] += sum_key [V^T ; 1][d][key] * P^T[key][q]
    if constexpr (P8) {
      // the lane's 32 probabilities of this tile = its 32 k-slots, already in slot order: register v = 4*k2 + g
      i32x8_t pf;
#pragma unroll
      for (int v = 0; v < 8; ++v)
        pf[v] = (int)pack_fp8x4(s[v >> 2][4 * (v & 3) + 0], s[v >> 2][4 * (v & 3) + 1], s[v >> 2][4 * (v & 3) + 2], s[v >> 2][4 * (v & 3) + 3]);
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const char* vr = Vs + (d * 32 + l31) * VROWB + hi * 32;
        const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(vr), v1 = *reinterpret_cast<const u32x4_t*>(vr + 16);
        const i32x8_t vf = {(int)v0.x, (int)v0.y, (int)v0.z, (int)v0.w, (int)v1.x, (int)v1.y, (int)v1.z, (int)v1.w};
        // cbsz = blgp = 0: both operands OCP e4m3; scales 0x7f = 2^0 in every E8M0 byte
        o[d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o[d], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
#pragma unroll
    for (int k2 = 0; k2 < NSUB; ++k2) {
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        u32x4_t pf;
        if constexpr (SZ == 4) {
          pf = u32x4_t{__float_as_uint(s[k2][4 * sl + 0]), __float_as_uint(s[k2][4 * sl + 1]),
                       __float_as_uint(s[k2][4 * sl + 2]), __float_as_uint(s[k2][4 * sl + 3])};
        } else {
          pf = u32x4_t{Op16<TM>::pack(s[k2][8 * sl + 0], s[k2][8 * sl + 1]), Op16<TM>::pack(s[k2][8 * sl + 2], s[k2][8 * sl + 3]),
                       Op16<TM>::pack(s[k2][8 * sl + 4], s[k2][8 * sl + 5]), Op16<TM>::pack(s[k2][8 * sl + 6], s[k2][8 * sl + 7])};
        }
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(Vs + (d * 32 + l31) * VROWB + k2 * 32 * SZ + sl * 32 + hi * 16);
          AMma<TM>::mma(o[d], vf, pf);
        }
      }
    }
    }
    }   // compute
    if (t + 1 < ntile) store_tile((t + 1) & 1, (t + 1) * KEYS);
    __syncthreads();
  }
  };

  constexpr int LB = HD / 32, LR = ((HD % 32) / 8) * 4;
  static_assert(HD % 32 == 0 || HD % 32 == 16, "ones row must sit at row 0 or 16 of its 32-row block");
  if (OPT && !g_attn_exact_only && !a.exact_only) {
    run(OptTag<true>{}, true);
    // the denominator of this lane's query (lane half 0 holds it) bounds every probability of the row: below 2^14 none of them
    // reached the fp16 range (the converts SATURATE under MODE.FP16_OVFL, so an overflow would not show up as inf), and it is
    // positive unless every probability flushed to zero -- otherwise the exact pass decides
    const float lq = o[LB][LR];
    const int bad = (hi == 0 && q < a.Lq && !(lq > 0.f && lq < 16384.f)) ? 1 : 0;
    const bool wave_bad = __any(bad) != 0;
    if (__syncthreads_or(bad)) {
      if (a.fallbacks && threadIdx.x == 0) atomicAdd(a.fallbacks, 1u);      // (this workgroup stages its tiles twice: counted, ns2vc_unet_attn_fallbacks)
      run(OptTag<false>{}, wave_bad);
    }
  } else {
    run(OptTag<false>{}, true);
  }

  // ---- normalise and store O[q][h*HD + d]; lane holds d = dt*32 + 8*g + 4*hi + i.  The denominator is row HD of the
  // accumulator: block HD/32, register ((HD%32)/8)*4, lane half 0 -> broadcast to both halves
  const auto lsw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[LB][LR]), __float_as_uint(o[LB][LR]), false, false);
  const float l_tot = __uint_as_float(lsw[0]);
  const float inv = 1.0f / l_tot;
  TM* op = reinterpret_cast<TM*>(a.out) + ((size_t)(b * a.Lq + min(q, a.Lq - 1)) * a.ldo + h * HD);
  if constexpr (SZ == 2) {
    // 16-bit results: the two lane halves of a query trade 4-element groups so that every lane stores 8 consecutive d (16 B) -- half the store
    // instructions of the 8-byte form, and the output of a workgroup is nothing but row-per-lane pieces whose issue sets its tail (r5: without
    // any output store the step ran 2.2 % faster)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        if (d * 32 + 16 * gp < HD) {
          const uint32_t a0 = Op16<TM>::pack(o[d][8 * gp] * inv, o[d][8 * gp + 1] * inv), a1 = Op16<TM>::pack(o[d][8 * gp + 2] * inv, o[d][8 * gp + 3] * inv);
          const uint32_t b0 = Op16<TM>::pack(o[d][8 * gp + 4] * inv, o[d][8 * gp + 5] * inv), b1 = Op16<TM>::pack(o[d][8 * gp + 6] * inv, o[d][8 * gp + 7] * inv);
          const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);      // lower half: group 2 gp of both halves; upper half: group 2 gp + 1
          const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
#if defined(NS2VC_ATTN_ABLATE_STORE)      // diagnostic build (wrong results, timing only): no output stores
          asm volatile("" :: "v"(x0[0]), "v"(x1[0]), "v"(x0[1]), "v"(x1[1]));
#else
          if (q < a.Lq) *reinterpret_cast<u32x4_t*>(op + d * 32 + 8 * (2 * gp + hi)) = u32x4_t{x0[0], x1[0], x0[1], x1[1]};
#endif
        }
      }
  } else if (q < a.Lq) {
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int d0 = d * 32 + 8 * gq + 4 * hi;
        if (d0 < HD) store_op4<TM>(op + d0, o[d][4 * gq] * inv, o[d][4 * gq + 1] * inv, o[d][4 * gq + 2] * inv, o[d][4 * gq + 3] * inv);
      }
  }
}

template <typename TM, int HD, int KEYS> static constexpr size_t attn_lds() {
  constexpr int SZ = AMma<TM>::SZ;
  constexpr int HDX = (HD + 1 + 31) / 32 * 32;
  return 2 * (size_t)(KEYS * (HD * SZ + 48) + HDX * (KEYS * SZ + 16));
}

// 64-key tiles everywhere.  128-key tiles (half the per-tile bookkeeping, barriers and waits; round-1 review item) exist for
// the narrow heads of the 16-bit precisions (hd 16 / 32) behind the tuning hook, and LOSE: same-box 3.93 (64) vs 3.98 ms/step
// (128), attention 0.68 vs 0.73 ms -- 64 more score registers and twice the LDS per workgroup cost more occupancy than the
// bookkeeping saves in a VALU-bound loop.
template <typename TM, int HD> static constexpr bool attn_has128() { return sizeof(TM) == 2 && HD <= 32; }
static int g_force_keys = 0;   // test / tuning hook (ns2vc_debug_set_attn_keys): 128 selects the 128-key kernels
void set_forced_attn_keys(int keys) { g_force_keys = keys; }
void set_attn_optimistic(int on) { const int v = on ? 0 : 1; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_exact_only), &v, sizeof(v)); }

template <typename TM, int HD, int KEYS> static hipError_t launch_hdk(const AttnArgs& a, hipStream_t s) {
  dim3 grid(((a.Lq + 127) / 128) * a.H * a.B);
  const size_t lds = attn_lds<TM, HD, KEYS>();          // (the fp8 variant needs less: its V^T rows are half as long)
  hipLaunchKernelGGL((attn_kernel<TM, HD, KEYS, false>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}
template <typename TM, int HD> static hipError_t launch_hd(const AttnArgs& a, hipStream_t s) {
  if constexpr (sizeof(TM) == 2) {
    if (a.pv_fp8) {
      dim3 grid(((a.Lq + 127) / 128) * a.H * a.B);
      const size_t lds = attn_lds<TM, HD, 64>();
      hipLaunchKernelGGL((attn_kernel<TM, HD, 64, true>), grid, dim3(256), lds, s, a);
      return hipGetLastError();
    }
  } else if (a.pv_fp8) return hipErrorInvalidValue;
  if constexpr (attn_has128<TM, HD>()) {
    if (g_force_keys == 128) return launch_hdk<TM, HD, 128>(a, s);
  }
  return launch_hdk<TM, HD, 64>(a, s);
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
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<TM, HD, 64, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)attn_lds<TM, HD, 64>());
  if constexpr (attn_has128<TM, HD>()) {
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<TM, HD, 128, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)attn_lds<TM, HD, 128>());
  }
  if constexpr (sizeof(TM) == 2) {
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<TM, HD, 64, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)attn_lds<TM, HD, 64>());
  }
  return e;
}
template <typename TM> static hipError_t set_attr_tm() {
  hipError_t e;
  if ((e = set_attr<TM, 16>()) != hipSuccess) return e;
  if ((e = set_attr<TM, 32>()) != hipSuccess) return e;
  if ((e = set_attr<TM, 48>()) != hipSuccess) return e;
  return set_attr<TM, 64>();
}
hipError_t init_attn_attributes() {
  hipError_t e = set_attr_tm<float>();
  if (e == hipSuccess) e = set_attr_tm<bf16_t>();
  if (e == hipSuccess) e = set_attr_tm<f16_t>();
  return e;
}

hipError_t launch_attention(const AttnArgs& a, int head_dim, int prec, hipStream_t s) {
  const int al = prec == PREC_F32 ? 3 : 7;       // rows must start 16-B aligned
  if (!a.q || !a.k || !a.v || !a.out) return hipErrorInvalidValue;
  if (a.Lq <= 0 || a.Lk <= 0 || (a.ldq & al) || (a.ldk & al) || (a.ldv & al) || (a.ldo & al)) return hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(a.out) & 15) return hipErrorInvalidValue;      // (results leave in 16-byte pieces)
  switch (prec) {
    case PREC_BF16: return launch_tm<bf16_t>(a, head_dim, s);
    case PREC_F16: return launch_tm<f16_t>(a, head_dim, s);
    case PREC_F32: return launch_tm<float>(a, head_dim, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace ns2vc
