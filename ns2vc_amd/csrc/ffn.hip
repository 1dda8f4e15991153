// Fused feed-forward + proj_out of a transformer block of the NS2VC denoiser, CDNA4 (gfx950), 16-bit operand types.
//
// Replaces, in ONE launch, the tail of BasicTransformerBlock / Transformer2DModel
// (reference unet1d/attention.py:178-203, 206-301 GEGLU with erf GELU; unet1d/transformer_1d.py:287-295):
//
//     n   = LayerNorm(y)                                   (norm3, gamma/beta folded into W1 / b1 at pack time)
//     h   = (n W1v^T + b1v) * gelu(n W1g^T + b1g)          (ff.net.0: GEGLU, hidden = 4 dim)
//     out = Wpo (y + W2 h + b2) + bpo + x                  (ff.net.2, + residual, proj_out (1x1 conv), + block residual)
//         = [Wpo W2 | Wpo] [h | y] + (Wpo b2 + bpo) + x    (folded at pack time: engine.cpp pack_all)
//
// As separate launches the GEGLU GEMM (K = dim: one or two K tiles, N = 8 dim) was the slowest GEMM of the step at every
// level (30-44 us: prologue, two cold K tiles and an erf epilogue per workgroup, 15 workgroups per CU, and a 30 MB
// hidden tensor written and read back), followed by a K = 5 dim GEMM.  Here one workgroup owns 64 tokens for the WHOLE
// chain: the raw rows y stay in LDS as the K panel of ff.net.0 AND of the Wpo segment, the hidden tensor never leaves
// registers, and the only stream from L2 is the weights -- one flat sequence of 16 KB tiles ([128 rows][64 k], already
// in the bank-conflict-free XOR-swizzled LDS image, in exactly the order the kernel consumes them), so the loader is a
// lane-linear LDS-DMA with every address in SGPRs.
//
// Everything is computed TRANSPOSED (the attention kernel's trick): S^T = W1 y^T, so after the 32x32 MFMA a lane holds
// 16 hidden units of ONE token (column = lane & 31): value and gate of a unit sit in the same lane and register, the
// LayerNorm fix-up is per lane (mean / rstd of the lane's token), GEGLU is pure register arithmetic, and the packed
// result IS the B operand of the second MFMA chain  O^T += W2' h^T  (W2' columns are stored with unit bits 2 and 3
// swapped so that the 8 units a lane half holds are contiguous in k).  512 threads = 8 waves = 2 token halves x 4
// hidden-unit groups: every wave accumulates O^T[all dim channels][its 32 tokens] over ITS hidden units; the four
// partials meet in the LDS-staged epilogue (bias, fp32 residual, fp32 + operand stores, GroupNorm statistics).
#include "common.h"
#include <type_traits>
#include "mma.h"
#include <vector>

namespace ns2vc {

typedef ::ns2vc_ffn_args FfnArgs;

// optional per-workgroup phase timing (cycles, wave 0): [block][8] = entry, prologue end, sum ff.net.0 steps, sum GEGLU,
// sum W2' steps, Wpo steps, epilogue, exit.  Compiled in only with -DNS2VC_GEMM_TRACE=1 (`make TRACE=1`), set through
// ns2vc_debug_set_gemm_trace; read by tools/ffn_trace.py
__device__ unsigned long long* g_ffn_trace = nullptr;
#ifndef NS2VC_GEMM_TRACE
#define NS2VC_GEMM_TRACE 0
#endif
// fragment reads vs MFMAs of a step: 1 = wait for ALL reads, then the MFMAs back to back (scheduling fence); 2 = leave the
// counted waits to the compiler but pin "reads first" with a scheduling barrier; 0 = compiler's choice
#ifndef NS2VC_FFN_FENCE
#define NS2VC_FFN_FENCE 1
#endif
#ifndef NS2VC_FFN_WARM
#define NS2VC_FFN_WARM 1
#endif
#if NS2VC_FFN_FENCE == 1
#define FFN_FRAG_FENCE() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#elif NS2VC_FFN_FENCE == 2
#define FFN_FRAG_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FFN_FRAG_FENCE() do {} while (0)
#endif
#if NS2VC_GEMM_TRACE
#define FFN_NOW() __builtin_readcyclecounter()
#define FFN_TR(i, v) do { if (tr && tid == 0) tr[i] = (v); } while (0)
#else
#define FFN_NOW() 0ull
#define FFN_TR(i, v) do { (void)tr; } while (0)
#endif
void set_ffn_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ffn_trace), &p, sizeof(p)); }

#ifndef NS2VC_XATT_ABLATE
#define NS2VC_XATT_ABLATE 0      // diagnostic builds of the in-kernel cross-attention (wrong results): 1 = no pass 1, 2 = no exponentials, 4 = no P V products, 8 = fragments loaded once
#endif
// helpers of the in-kernel cross-attention: combine a value with lane ^ 32; 2^x on v_exp_f32
__device__ __forceinline__ float half_max_f(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float exp2f_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// tile stream geometry (host packer below and kernel must agree)
constexpr int FFN_TILE = 128 * 128;        // bytes: [128 rows][128 B of K]
constexpr int FFN_PAIR = 2 * FFN_TILE;     // the kernel consumes tiles two at a time

template <int D, bool PRE = false> struct FfnGeom {
  // pairs resident in LDS.  The stream is LATENCY-bound (one workgroup per CU, ~2 us per piece under load: bytes in flight
  // per CU set the rate), so the ring is as deep as the 160 KB allow: 4 pairs at dim 128 (3 in flight), 3 at dim 256
  static constexpr int RING = D <= 128 ? 4 : 3;
  static constexpr int KT = D / 64;        // K tiles of ff.net.0 (= tiles of the token panel)
  static constexpr int NB = D / 32;        // 32-channel output blocks (accumulator tiles per wave)
  static constexpr int NSS = D / 32;       // super-steps of 128 hidden units (4 D hidden in all)
  static constexpr int NB128 = D / 128;    // 128-row blocks of W2' / Wpo
  static constexpr int PO_STEPS = NB128 * KT / 2;
  static constexpr int PRE_PAIRS = PRE ? NB128 * KT / 2 : 0;     // attn2.to_out (dim x dim) in front, k-tile outer / row block inner
  static constexpr int PAIRS = PRE_PAIRS + NSS * (KT + NB128) + PO_STEPS;
  static constexpr int PANEL = KT * 64 * 128;               // bytes: 64 tokens x D, as KT swizzled [64][128 B] tiles
  static constexpr int CONSTS = 8 * D * 8;                  // bytes: (rowsum, bias) per packed W1 row
  static constexpr int XCH = PRE ? 64 * 4 * 8 : 0;          // bytes: (sum, sumsq) per token and row group of the pre-stage result
  static constexpr int LDS = RING * FFN_PAIR + PANEL + CONSTS + XCH;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static constexpr int EP = 36;                             // epilogue staging pitch (floats)
  static_assert(2 * 2 * 4 * 32 * EP * 4 <= RING * FFN_PAIR, "epilogue staging (two 32-channel blocks) fits in the ring");
  static_assert(NB * 512 * 8 <= PANEL, "per-lane statistics partials fit in the panel");
};

// ATT (with PRE): the prompt cross-attention of attn2 computed here too (r6, FfnArgs.att_*): the panel the pre-stage multiplies is PRODUCED by the eight
// waves -- one per head -- instead of being fetched from an attention launch's output.
template <typename TM, int D, bool PRE, bool ATT = false>
__global__ __launch_bounds__(512) void ffn_kernel(const FfnArgs a) {
  op_mode_init<TM>();
  static_assert(!ATT || PRE, "the in-kernel cross-attention feeds the pre-stage");
  using G = FfnGeom<D, PRE>;
  constexpr int KT = G::KT, NB = G::NB, NSS = G::NSS, NB128 = G::NB128, NP = G::PAIRS, EP = G::EP, RING = G::RING;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const panel = smem + RING * FFN_PAIR;
  const float* const consts = reinterpret_cast<const float*>(panel + G::PANEL);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tw = wave & 1, hw = wave >> 1;          // token half, hidden-unit group
  const int l31 = lane & 31, hi = lane >> 5;
  const int sw = (l31 >> 1) & 7;                    // read-side XOR swizzle of every fragment row this lane touches
  const unsigned lds0 = (unsigned)(size_t)smem;
  // token block: 64 consecutive rows of the B*T tokens; with the in-kernel cross-attention the blocks are cut per batch item (a block's queries share
  // one item's keys): ceil(T / 64) workgroups per item, rows past the item's end are masked like rows past M
  const int nbi = ATT ? (a.T + 63) >> 6 : 1;
  const int ab = ATT ? (int)blockIdx.x / nbi : 0;
  const int m0 = ATT ? ab * a.T + ((int)blockIdx.x - ab * nbi) * 64 : (int)blockIdx.x * 64;
  const int mlim = ATT ? (ab + 1) * a.T : a.M;      // first row that is not this block's any more
  const int mtok = m0 + 32 * tw + l31;              // this lane's token (both lane halves)
  unsigned long long* const tr = (NS2VC_GEMM_TRACE && g_ffn_trace) ? g_ffn_trace + (size_t)blockIdx.x * 8 : nullptr;
  unsigned long long t_ff1 = 0, t_gg = 0, t_ff2 = 0, t_mark = 0;
  (void)t_ff1; (void)t_gg; (void)t_ff2;             // (only read in the trace build)
  FFN_TR(0, FFN_NOW());

  const i32x4_t rW = make_rsrc(a.wstream, (unsigned long long)NP * FFN_PAIR);
  const unsigned lane16 = (unsigned)(lane * 16);
  if constexpr (ATT) {
    // ---- prompt cross-attention of this block's 64 tokens, one wave per head (r6).  The k | v of the prompt are hoisted per utterance, so the SDPA of
    // attn2 is token-local: S^T = K Q^T per 32 keys x 32 queries (a lane then holds 16 keys of ONE query: the row maximum is lane-local plus one exchange
    // with lane ^ 32), two passes over the keys -- maximum first, then p = exp2(s - max) rounded to the operand type and O^T += V^T P^T -- no running
    // rescale, no LDS, no barrier: K and V^T fragments come from the per-utterance fragment image (ns2vc_k_xattn_pack), 1 KB of consecutive bytes per wave
    // and load (a first build read the k rows in place: 32 cache lines of 17 KB stride per load for 1 KB of data -- the phase ran at the L2 request rate).  The mask bias (and -inf for the keys past Lk) enters through one more MFMA k-slab: K_aux = {bias / scale},
    // Q_aux = {1}.  The softmax scale is applied in fp32 (fma(s, scale*log2e, -max)), the denominator is the sum of the ROUNDED probabilities (hd 16: the
    // ones row of V^T inside the P V product; hd 32: a product with a constant ones fragment).
    constexpr int HD = D / 8, NSL = HD / 16;
    const int Lk = a.att_Lk, Lpad = (Lk + 31) & ~31, ntile = Lpad >> 5;
    const int hd0 = wave * HD;                                   // this wave's head = its first channel
    // this (item, head)'s fragment image: per tile NSL K fragments + 2 V^T fragments of 1 KB, lane-linear (ns2vc_k_xattn_pack)
    const char* const KV = reinterpret_cast<const char*>(a.att_kv) + ((size_t)(ab * 8 + wave) * ntile) * ((NSL + 2) * 1024) + lane * 16;
    const float* const bias = a.att_bias ? a.att_bias + (size_t)ab * Lk : nullptr;
    const float sc2 = a.att_scale * 1.4426950408889634f, rsc = 1.0f / a.att_scale;
    const u32x4_t qaux = hi == 0 ? u32x4_t{Op16<TM>::pack(1.0f, 0.f), 0u, 0u, 0u} : u32x4_t{0u, 0u, 0u, 0u};
    const u32x4_t ones = l31 == 0 ? u32x4_t{Op16<TM>::pack(1.0f, 1.0f), Op16<TM>::pack(1.0f, 1.0f), Op16<TM>::pack(1.0f, 1.0f), Op16<TM>::pack(1.0f, 1.0f)}
                                  : u32x4_t{0u, 0u, 0u, 0u};
    (void)ones;
    // One pass over the key tiles serves BOTH 32-query halves of the block (the K / V^T fragments are loaded once), and the fragments are requested PD tiles
    // ahead: a tile's arithmetic is ~500 cycles, a fragment load from L2 with every CU streaming 1-2 k (first build: one tile ahead, one half at a time --
    // 70 us per launch, the loop ran at load latency).
    constexpr int PD = 3;
    auto load_k = [&](int t, u32x4_t (&kf)[NSL], float& kb) __attribute__((always_inline)) {
      // (branch-free: tiles past the end re-read the last one and get -inf for every key, so the loop body below has no predicates around its loads and the
      //  compiler can count them -- with `if (t < ntile)` around the loads it fell back to vmcnt(0) before every tile: no prefetch at all)
      const int key = 32 * t + l31;
      const char* kp = KV + (size_t)min(t, ntile - 1) * ((NSL + 2) * 1024);
#pragma unroll
      for (int s2 = 0; s2 < NSL; ++s2) kf[s2] = *reinterpret_cast<const u32x4_t*>(kp + s2 * 1024);
      const float bvv = bias ? bias[min(key, Lk - 1)] * rsc : 0.f;
      kb = key < Lk ? bvv : -__builtin_inff();
    };
    auto scores = [&](const u32x4_t (&kf)[NSL], float kb, const u32x4_t (&qf)[NSL]) __attribute__((always_inline)) {
      const u32x4_t kaux = hi == 0 ? u32x4_t{Op16<TM>::pack(kb, 0.f), 0u, 0u, 0u} : u32x4_t{0u, 0u, 0u, 0u};
      f32x16_t sacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
      MmaT<TM>::mma(sacc, kaux, qaux);
#pragma unroll
      for (int s2 = 0; s2 < NSL; ++s2) MmaT<TM>::mma(sacc, kf[s2], qf[s2]);
      return sacc;
    };
    u32x4_t qf[2][NSL];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      const TM* qp = reinterpret_cast<const TM*>(a.att_q) + (size_t)min(m0 + 32 * qs + l31, mlim - 1) * a.att_ldq + hd0 + 8 * hi;
#pragma unroll
      for (int s2 = 0; s2 < NSL; ++s2) qf[qs][s2] = *reinterpret_cast<const u32x4_t*>(qp + 16 * s2);
    }
    // pass 1: the row maxima (raw scores; the scale is positive)
    float mx[2] = {-__builtin_inff(), -__builtin_inff()};
    {
      u32x4_t kf[PD][NSL];
      float kb[PD];
#pragma unroll
      for (int j = 0; j < PD; ++j) load_k(j, kf[j], kb[j]);
#pragma unroll 1
      for (int t0 = 0; t0 < ((NS2VC_XATT_ABLATE & 1) ? 0 : ntile); t0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
#pragma unroll
          for (int qs = 0; qs < 2; ++qs) {
            const f32x16_t sa = scores(kf[j], kb[j], qf[qs]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx[qs] = fmaxf(mx[qs], fmaxf(sa[r], sa[r + 1]));
          }
          if (!(NS2VC_XATT_ABLATE & 8)) load_k(t0 + j + PD, kf[j], kb[j]);
        }
      }
    }
    float m2[2];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) m2[qs] = half_max_f(mx[qs]) * sc2;
    // pass 2: probabilities and P V
    constexpr int NO = HD == 16 ? 1 : 2;
    f32x16_t oacc[2][NO];
#pragma unroll
    for (int qs = 0; qs < 2; ++qs)
#pragma unroll
      for (int i = 0; i < NO; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[qs][i][r] = 0.f;
    {
      u32x4_t kf[PD][NSL], vf[PD][2];
      float kb[PD];
      auto load_v = [&](int t, u32x4_t (&v)[2]) __attribute__((always_inline)) {
        const char* vp = KV + (size_t)min(t, ntile - 1) * ((NSL + 2) * 1024) + NSL * 1024;
        v[0] = *reinterpret_cast<const u32x4_t*>(vp); v[1] = *reinterpret_cast<const u32x4_t*>(vp + 1024);
      };
#pragma unroll
      for (int j = 0; j < PD; ++j) { load_k(j, kf[j], kb[j]); load_v(j, vf[j]); }
#pragma unroll 1
      for (int t0 = 0; t0 < ntile; t0 += PD) {
#pragma unroll
        for (int j = 0; j < PD; ++j) {
          {
#pragma unroll
            for (int qs = 0; qs < 2; ++qs) {
              const f32x16_t sa = scores(kf[j], kb[j], qf[qs]);
              float pr[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) pr[r] = (NS2VC_XATT_ABLATE & 2) ? sa[r] : exp2f_fast(fmaf(sa[r], sc2, -m2[qs]));
              const u32x4_t p0 = u32x4_t{Op16<TM>::pack(pr[0], pr[1]), Op16<TM>::pack(pr[2], pr[3]), Op16<TM>::pack(pr[4], pr[5]), Op16<TM>::pack(pr[6], pr[7])};
              const u32x4_t p1 = u32x4_t{Op16<TM>::pack(pr[8], pr[9]), Op16<TM>::pack(pr[10], pr[11]), Op16<TM>::pack(pr[12], pr[13]), Op16<TM>::pack(pr[14], pr[15])};
              if (NS2VC_XATT_ABLATE & 4) { oacc[qs][0][0] += __uint_as_float(p0.x ^ p1.y ^ vf[j][0].x ^ vf[j][1].w); }
              else {
              MmaT<TM>::mma(oacc[qs][0], vf[j][0], p0);
              MmaT<TM>::mma(oacc[qs][0], vf[j][1], p1);
              if constexpr (HD == 32) { MmaT<TM>::mma(oacc[qs][1], ones, p0); MmaT<TM>::mma(oacc[qs][1], ones, p1); }
              }
            }
            if (!(NS2VC_XATT_ABLATE & 8)) { load_k(t0 + j + PD, kf[j], kb[j]); load_v(t0 + j + PD, vf[j]); }
          }
        }
      }
    }
    // normalise and write this head's columns of the panel: lane (q, hi) holds d = 8 (r >> 2) + 4 hi + (r & 3)
#pragma unroll
    for (int qs = 0; qs < 2; ++qs) {
      const int tokq = 32 * qs + l31;
      float den = HD == 16 ? oacc[qs][0][8] : oacc[qs][NO - 1][0];     // (row 16 = the ones row / row 0 of the ones product: in the hi = 0 lanes)
      den = __shfl(den, l31);
      const float inv = m0 + tokq < mlim ? 1.0f / den : 0.f;
      const int swt = (tokq >> 1) & 7;
#pragma unroll
      for (int g = 0; g < HD / 8; ++g) {
        const int n = hd0 + 8 * g + 4 * hi;
        char* dst = panel + (n >> 6) * 8192 + tokq * 128 + ((((n & 63) >> 3) ^ swt) * 16) + 8 * hi;
        *reinterpret_cast<uint2*>(dst) = make_uint2(Op16<TM>::pack(oacc[qs][0][4 * g] * inv, oacc[qs][0][4 * g + 1] * inv),
                                                    Op16<TM>::pack(oacc[qs][0][4 * g + 2] * inv, oacc[qs][0][4 * g + 3] * inv));
      }
    }
    // (the panel is complete for everybody at the first step's barrier: step_begin waits for this wave's LDS writes and meets the others.  The phase runs
    //  BEFORE any LDS-DMA is issued: its loads are ordinary ones, whose compiler-placed waits would otherwise also cover every DMA issued before them)
  }
#if NS2VC_FFN_WARM
  // ---- L2 warm-up.  In the captured step this layer's weights are cold (every layer's weights are read once per step and
  // 132 MB of them pass through the 4 MB L2s in between), every workgroup consumes the SAME tiles at the same pace, and a
  // cold pair takes ~2-4 us to arrive: with two pairs in flight the stream crawled at one HBM latency per pair (52 pairs =
  // 52 us at dim 256, whatever the kernel did in between).  So before anything else the workgroups of an XCD pull
  // DISJOINT slices of the stream through that XCD's L2 (block b runs on XCD b % 8 -- observed placement, only speed depends
  // on it): LDS-DMA into the last ring slot (unused until the first refill, which is issued later and therefore lands later).
  {
    const unsigned nx = (gridDim.x + 7) >> 3, jx = blockIdx.x >> 3;
    const unsigned total = (unsigned)NP * FFN_PAIR;
    const unsigned per = (((total + nx - 1) / nx) + 8191u) & ~8191u;
    for (unsigned off = jx * per + wave * 1024; off < min(total, (jx + 1) * per); off += 8192)
      blds16(rW, lane16, off, lds0 + (RING - 1) * FFN_PAIR + wave * 1024);
  }
#endif
  // ---- DMA: token panel (source-side swizzle, rows past M read as zeros), constants, then the weight stream
  {
    const int prow = 8 * wave + (lane >> 3), pchunk = lane & 7;           // one 1-KB piece per wave = 8 rows x 128 B
    const int m = m0 + prow;
    const int ldp = PRE ? a.pre_lda : a.ldy;          // PRE: the panel starts as the attention output rows (pre-stage A operand)
    const unsigned voff = m < mlim ? (unsigned)m * (unsigned)ldp * 2u + (unsigned)((pchunk ^ ((prow >> 1) & 7)) * 16) : DMA_OOB;
    const i32x4_t rY = make_rsrc(PRE ? a.pre_a : a.yn, (unsigned long long)a.M * ldp * 2ull);
    if constexpr (!ATT) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) blds16(rY, voff, (unsigned)(kt * 128), lds0 + RING * FFN_PAIR + kt * 8192 + wave * 1024);
    }
    const i32x4_t rC = make_rsrc(a.consts, (unsigned long long)G::CONSTS);
#pragma unroll
    for (int j = 0; j < G::CONSTS / 8192; ++j)
      blds16(rC, (unsigned)(lane * 16), (unsigned)((j * 8 + wave) * 1024), lds0 + RING * FFN_PAIR + G::PANEL + (j * 8 + wave) * 1024);
  }
  // pre-stage bias / residual rows of this lane's token: issued BEFORE the weight pairs so that the counted waits on the
  // pairs (loads complete in issue order) do not have to sit through them
  float4 b0[NB128][4], rr0[NB128][4];
  if constexpr (PRE) {
#pragma unroll
    for (int rb = 0; rb < NB128; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 128 * rb + 32 * hw + 8 * g + 4 * hi;
        b0[rb][g] = *reinterpret_cast<const float4*>(a.pre_bias + n);
        rr0[rb][g] = mtok < mlim ? *reinterpret_cast<const float4*>(a.pre_res + (size_t)mtok * a.pre_ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  // a pair = 32 pieces of 1 KB: four per wave, all addresses scalar
  auto issue_piece = [&](int p, int j) __attribute__((always_inline)) {
    const int slot = p % RING;
    blds16(rW, lane16, (unsigned)(p * FFN_PAIR + (j * 8 + wave) * 1024), lds0 + slot * FFN_PAIR + (j * 8 + wave) * 1024);
  };
  auto issue_pair = [&](int p) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_piece(p, j);
  };
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < NP) issue_pair(q);

  // ---- LayerNorm statistics of this lane's token (ordinary loads: the compiler waits for them -- and, not seeing the
  // DMA above, for everything issued so far: that is the prologue's wait for the first tiles anyway).  With the pre-stage
  // they come out of it (below) instead of from memory.
  float mean = 0.f, rstd = 1.f;
  auto ln_finish = [&](float s, float q) __attribute__((always_inline)) {
    const float inv = 1.0f / (float)D;
    mean = s * inv;
    double var = (double)q * (double)inv - (double)mean * (double)mean;
    if (var < 0.0) var = 0.0;
    rstd = 1.0f / sqrtf((float)var + a.ln_eps);
    if (a.ln_health && hw == 0) {          // same health report as the LayerNorm-consumer GEMMs (gemm.hip ln_row_finish)
      float ratio = mtok < mlim ? fabsf(mean) * rstd : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
      if (lane == 0 && ratio > __uint_as_float(__hip_atomic_load(a.ln_health, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(a.ln_health, __float_as_uint(ratio));
    }
  };
  if constexpr (!PRE) {
    const float4* sp = reinterpret_cast<const float4*>(a.ln_stats + (size_t)min(mtok, a.M - 1) * (D / 64) * 2);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < D / 128; ++i) { const float4 v = sp[i]; s += v.x + v.z; q += v.y + v.w; }
    ln_finish(s, q);
  }

  f32x16_t accO[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accO[i][r] = 0.f;
  FFN_TR(1, FFN_NOW());

  int p = 0;                                        // next pair to consume
  auto step_begin = [&](bool first = false) __attribute__((always_inline)) -> const char* {
    // pair p has landed when only the pieces (four per wave and pair) of the RING - 2 pairs behind it may still be in flight
    const int after = min(RING - 2, NP - 1 - p);
    if (RING >= 4 && after >= 2) wait_vmcnt<8>();
    else if (after >= 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads of the slot about to be refilled are done
    __builtin_amdgcn_s_barrier();
    // refill: pair p + RING - 1 into the slot of pair p-1, free for everyone after the barrier.  (Issuing the four
    // pieces one by one between the MFMA groups instead was measured slower: 4.44 vs 4.32 ms/step same-box.)
    if (PRE && first) {
      // the compiler does not see the DMA: consume the pre-stage bias / residual registers HERE, before any younger memory
      // operation exists -- otherwise its `s_waitcnt vmcnt(0)` at their first use also covers the prefetched pairs (see
      // rowchain.hip step_begin)
#pragma unroll
      for (int rb = 0; rb < NB128; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          asm volatile("" : "+v"(b0[rb][g].x), "+v"(b0[rb][g].y), "+v"(b0[rb][g].z), "+v"(b0[rb][g].w));
          asm volatile("" : "+v"(rr0[rb][g].x), "+v"(rr0[rb][g].y), "+v"(rr0[rb][g].z), "+v"(rr0[rb][g].w));
        }
    }
    if (p + RING - 1 < NP) issue_pair(p + RING - 1);
    return ring + (p % RING) * FFN_PAIR;
  };
  const char* const bpanel = panel + (32 * tw + l31) * 128;    // this lane's token row inside a panel tile (+ kt * 8192)

  if constexpr (PRE) {
    // ---- pre-stage (attn2.to_out + residual, attention_processor.py:1040-1050, attention.py:160): y^T = Wo o^T + bo + y_prev^T.
    // Wave (tw, hw) takes rows 32 hw .. + 31 of every 128-row tile for its 32 tokens; y stays on the CU: its operand copy
    // replaces o in the panel (K panel of ff.net.0 and of the Wpo segment), its row sums give this lane's mean / rstd.
    f32x16_t acc0[NB128];
#pragma unroll
    for (int i = 0; i < NB128; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < G::PRE_PAIRS; ++j) {
      const char* T = step_begin(j == 0);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int i = 2 * j + t, kt = i / NB128, rb = i % NB128;
        const char* wrow = T + t * FFN_TILE + (32 * hw + l31) * 128;
        u32x4_t fb[4], fw[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int c = ((2 * ks + hi) ^ sw) * 16;
          fb[ks] = *reinterpret_cast<const u32x4_t*>(bpanel + kt * 8192 + c);
          fw[ks] = *reinterpret_cast<const u32x4_t*>(wrow + c);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) MmaT<TM>::mma(acc0[rb], fw[ks], fb[ks]);
      }
      ++p;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // every wave is done reading o: the panel may be overwritten with y
    float2* const xch = reinterpret_cast<float2*>(panel + G::PANEL + G::CONSTS);
    const int tok = 32 * tw + l31;
    float ps = 0.f, pq = 0.f;
#pragma unroll
    for (int rb = 0; rb < NB128; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = acc0[rb][4 * g + 0] + b0[rb][g].x + rr0[rb][g].x;
        v.y = acc0[rb][4 * g + 1] + b0[rb][g].y + rr0[rb][g].y;
        v.z = acc0[rb][4 * g + 2] + b0[rb][g].z + rr0[rb][g].z;
        v.w = acc0[rb][4 * g + 3] + b0[rb][g].w + rr0[rb][g].w;
        if (mtok >= mlim) v = make_float4(0.f, 0.f, 0.f, 0.f);
        ps += (v.x + v.y) + (v.z + v.w);
        pq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        const int n = 128 * rb + 32 * hw + 8 * g + 4 * hi;
        char* dst = panel + (n >> 6) * 8192 + tok * 128 + ((((n & 63) >> 3) ^ sw) * 16) + 8 * hi;
        *reinterpret_cast<uint2*>(dst) = make_uint2(Op16<TM>::pack(v.x, v.y), Op16<TM>::pack(v.z, v.w));
      }
    {
      const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
      const auto q2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pq), __float_as_uint(pq), false, false);
      ps = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
      pq = __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
      if (hi == 0) xch[tok * 4 + hw] = make_float2(ps, pq);
    }
    __syncthreads();                                // y panel and row sums complete
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float2 v = xch[tok * 4 + k]; s += v.x; q += v.y; }     // fixed order: deterministic
    ln_finish(s, q);
  }

#pragma unroll 1
  for (int ss = 0; ss < NSS; ++ss) {
    // ---- ff.net.0 (transposed): value / gate pre-activations of this wave's 32 hidden units for its 32 tokens
    f32x16_t av, ag;
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; }
    t_mark = FFN_NOW();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const char* T = step_begin();
      const char* wrow = T + (hw >> 1) * FFN_TILE + (64 * (hw & 1) + l31) * 128;      // value row; gate row = + 32 rows
      // all twelve fragment reads of the step are issued before its first MFMA (left to itself the compiler issues two
      // reads per MFMA and waits for them: ~180 exposed cycles of LDS latency per MFMA, 1700 cycles per step at dim 256)
      u32x4_t fb[4], fv[4], fg[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ((2 * ks + hi) ^ sw) * 16;
        fb[ks] = *reinterpret_cast<const u32x4_t*>(bpanel + kt * 8192 + c);
        fv[ks] = *reinterpret_cast<const u32x4_t*>(wrow + c);
        fg[ks] = *reinterpret_cast<const u32x4_t*>(wrow + 32 * 128 + c);
      }
      FFN_FRAG_FENCE();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        MmaT<TM>::mma(av, fv[ks], fb[ks]);
        MmaT<TM>::mma(ag, fg[ks], fb[ks]);
      }
      ++p;
    }
    // ---- LayerNorm fix-up + bias + GEGLU in registers; register r <-> unit (r&3) + 8 (r>>2) + 4 hi of this wave's 32
    if (NS2VC_GEMM_TRACE) { asm volatile("" : "+v"(av), "+v"(ag)); const unsigned long long t = FFN_NOW(); t_ff1 += t - t_mark; t_mark = t; }
    u32x4_t hf[2];
    {
      const float* cv = consts + (size_t)(256 * ss + 64 * hw) * 2;           // (rowsum, bias) of the value rows; gate rows = + 32
      float h[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v0 = *reinterpret_cast<const float4*>(cv + (8 * j + 4 * hi) * 2);
        const float4 v1 = *reinterpret_cast<const float4*>(cv + (8 * j + 4 * hi) * 2 + 4);
        const float4 g0 = *reinterpret_cast<const float4*>(cv + (32 + 8 * j + 4 * hi) * 2);
        const float4 g1 = *reinterpret_cast<const float4*>(cv + (32 + 8 * j + 4 * hi) * 2 + 4);
        const float wsv[4] = {v0.x, v0.z, v1.x, v1.z}, bv[4] = {v0.y, v0.w, v1.y, v1.w};
        const float wsg[4] = {g0.x, g0.z, g1.x, g1.z}, bg[4] = {g0.y, g0.w, g1.y, g1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pv = rstd * (av[4 * j + i] - mean * wsv[i]) + bv[i];
          const float pg = rstd * (ag[4 * j + i] - mean * wsg[i]) + bg[i];
          h[4 * j + i] = pv * gelu_erf_f(pg);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
        hf[s] = u32x4_t{Op16<TM>::pack(h[8 * s + 0], h[8 * s + 1]), Op16<TM>::pack(h[8 * s + 2], h[8 * s + 3]),
                        Op16<TM>::pack(h[8 * s + 4], h[8 * s + 5]), Op16<TM>::pack(h[8 * s + 6], h[8 * s + 7])};
    }
    // ---- O^T[n][token] += W2'[n][this wave's 32 units] h^T : 32 units = half of a 64-unit tile
    if (NS2VC_GEMM_TRACE) { asm volatile("" : "+v"(hf[0]), "+v"(hf[1])); const unsigned long long t = FFN_NOW(); t_gg += t - t_mark; t_mark = t; }
#pragma unroll
    for (int nb = 0; nb < NB128; ++nb) {
      const char* T = step_begin();
      const char* wrow = T + (hw >> 1) * FFN_TILE + l31 * 128;
      u32x4_t fw[2][4];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          fw[s][i] = *reinterpret_cast<const u32x4_t*>(wrow + i * 32 * 128 + ((4 * (hw & 1) + 2 * s + hi) ^ sw) * 16);
      FFN_FRAG_FENCE();
#pragma unroll
      for (int s = 0; s < 2; ++s)            // (s outer: four independent accumulators between two MFMAs on the same one)
#pragma unroll
        for (int i = 0; i < 4; ++i) MmaT<TM>::mma(accO[4 * nb + i], fw[s][i], hf[s]);
      ++p;
    }
    if (NS2VC_GEMM_TRACE) { asm volatile("" : "+v"(accO[0])); const unsigned long long t = FFN_NOW(); t_ff2 += t - t_mark; }
  }
  // ---- the Wpo segment: O^T += Wpo y^T; wave hw takes k-slab hw of every 64-wide K tile
  FFN_TR(2, t_ff1); FFN_TR(3, t_gg); FFN_TR(4, t_ff2);
  t_mark = FFN_NOW();
#pragma unroll
  for (int j = 0; j < G::PO_STEPS; ++j) {
    const char* T = step_begin();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int q = 2 * j + t, nb = q / KT, kt = q % KT;                     // (compile-time after unrolling)
      const int c = ((2 * hw + hi) ^ sw) * 16;
      const u32x4_t b = *reinterpret_cast<const u32x4_t*>(bpanel + kt * 8192 + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4_t w = *reinterpret_cast<const u32x4_t*>(T + t * FFN_TILE + (32 * i + l31) * 128 + c);
        MmaT<TM>::mma(accO[4 * nb + i], w, b);
      }
    }
    ++p;
  }

  // ---- epilogue: the four hidden-group partials of a token half meet in LDS (re-using the ring); 512 lanes = 2 halves x
  // 32 tokens x 8 channel quads take one float4 each per 32-channel block: bias, residual, stores, GroupNorm statistics
  __syncthreads();
  if (NS2VC_GEMM_TRACE) { const unsigned long long t = FFN_NOW(); FFN_TR(5, t - t_mark); t_mark = t; }
  float* const E = reinterpret_cast<float*>(ring);        // [2 blocks][token half][hidden group][32 tokens][EP] partial tiles
  float2* const P = reinterpret_cast<float2*>(panel);     // [NB][512 lanes] (sum, sum of squares) of each lane's float4 (panel is free now)
  constexpr int EBLK = 2 * 4 * 32 * EP;                   // floats per staged 32-channel block
  const int L = tid, eth = L >> 8, etok = (L >> 3) & 31, equad = L & 7;
  const int em = m0 + 32 * eth + etok;
  const bool eok = em < mlim;
  float4 rr[NB];                                          // residual rows + bias: every load in flight before the first store
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    rr[nb] = eok ? *reinterpret_cast<const float4*>(a.res + (size_t)em * a.ldres + 32 * nb + 4 * equad) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 bb = *reinterpret_cast<const float4*>(a.bias2 + 32 * nb + 4 * equad);
    rr[nb].x += bb.x; rr[nb].y += bb.y; rr[nb].z += bb.z; rr[nb].w += bb.w;
  }
  float* const Ew = E + ((tw * 4 + hw) * 32 + l31) * EP + 4 * hi;
  const float* const Er = E + (eth * 4 * 32 + etok) * EP + 4 * equad;
  TM* const oo = reinterpret_cast<TM*>(a.out_op);
#pragma unroll
  for (int r2 = 0; r2 < NB / 2; ++r2) {                   // two 32-channel blocks per barrier pair
    if (r2) lds_barrier();          // (LDS-only: the previous round's result stores stay in flight)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(Ew + u * EBLK + 8 * j) =
            make_float4(accO[2 * r2 + u][4 * j], accO[2 * r2 + u][4 * j + 1], accO[2 * r2 + u][4 * j + 2], accO[2 * r2 + u][4 * j + 3]);
    lds_barrier();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int nb = 2 * r2 + u;
      float4 v = *reinterpret_cast<const float4*>(Er + u * EBLK);
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(Er + u * EBLK + k * 32 * EP);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
      const int n = 32 * nb + 4 * equad;
      float ps = 0.f, pq = 0.f;
      if (eok) {
        v.x += rr[nb].x; v.y += rr[nb].y; v.z += rr[nb].z; v.w += rr[nb].w;
        if (a.out_f32) out_f4(a.out_f32 + (size_t)em * a.ldo_f32 + n, v.x, v.y, v.z, v.w);
        if (oo) out_op4<TM>(oo + (size_t)em * a.ldo_op + n, v.x, v.y, v.z, v.w);
        ps = (v.x + v.y) + (v.z + v.w); pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
      if (a.stats) P[nb * 512 + L] = make_float2(ps, pq);
    }
  }
  if (a.stats) {
    // GroupNorm statistics of the result: per 16-channel block, the 64 tokens x 4 channel quads are summed in a FIXED order
    // (lane partials above -> per-token sums -> shuffle tree over the tokens) and leave as ONE int64 fixed-point atomic
    // per (batch item, block, moment): deterministic, 8 .. 64 atomics per workgroup
    lds_barrier();
    constexpr int TPB = 512 / (2 * NB);                   // threads per 16-channel block: 64 (dim 128) / 32 (dim 256)
    constexpr int TPT = 64 / TPB;                         // tokens per thread
    const int blk = tid / TPB, tl = tid % TPB;
    const int b0 = min(m0, a.M - 1) / a.T;                // first batch item of this 64-token block (T >= 64: at most two)
    const int mB = (b0 + 1) * a.T;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
      const int tok = tl * TPT + k;                       // 0..63 inside the block of tokens
      const float2* pp = P + (blk >> 1) * 512 + (tok >> 5) * 256 + (tok & 31) * 8 + 4 * (blk & 1);
      float ss = 0.f, qq = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) { const float2 v = pp[c]; ss += v.x; qq += v.y; }
      if (m0 + tok < mB) { s0 += ss; q0 += qq; } else { s1 += ss; q1 += qq; }
    }
#pragma unroll
    for (int o = 1; o < TPB; o <<= 1) { s0 += __shfl_xor(s0, o); q0 += __shfl_xor(q0, o); s1 += __shfl_xor(s1, o); q1 += __shfl_xor(q1, o); }
    if (tl == 0) {
      unsigned long long* st = reinterpret_cast<unsigned long long*>(a.stats) + ((size_t)b0 * (D / 16) + blk) * 2;
      atomicAdd(st, (unsigned long long)llrint((double)s0 * GN_SUM_SCALE));
      atomicAdd(st + 1, (unsigned long long)llrint((double)q0 * GN_SQ_SCALE));
      if (mB < a.M && mB < m0 + 64) {
        atomicAdd(st + 2 * (D / 16), (unsigned long long)llrint((double)s1 * GN_SUM_SCALE));
        atomicAdd(st + 2 * (D / 16) + 1, (unsigned long long)llrint((double)q1 * GN_SQ_SCALE));
      }
    }
  }
  if (NS2VC_GEMM_TRACE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t = FFN_NOW(); (void)t; FFN_TR(6, t - t_mark); FFN_TR(7, t); }
}

// ---------------------------------------------------------------------------
// host side: tile-stream packer and launcher
// ---------------------------------------------------------------------------
// one 16 KB tile [128 rows][64 k] of a row-major fp32 matrix, rounded to the operand type, in the swizzled LDS image:
// byte r*128 + pos*16 holds logical chunk pos ^ ((r>>1)&7) of row r; `perm` (optional) maps a stored k to the source column
static void append_tile(std::vector<unsigned short>& out, const float* mat, size_t ld, int row0, int col0, int prec, const int* perm) {
  for (int r = 0; r < 128; ++r)
    for (int pos = 0; pos < 8; ++pos) {
      const int lc = pos ^ ((r >> 1) & 7);
      for (int e = 0; e < 8; ++e) {
        const int k = lc * 8 + e;
        const int col = col0 + (perm ? perm[k] : k);
        out.push_back(f32_to_op16_bits(mat[(size_t)(row0 + r) * ld + col], prec));
      }
    }
}

// w0 (optional): the pre-stage matrix [dim][dim] (attn2.to_out), row-major fp32; its tiles go in front, k-tile outer
hipError_t pack_ffn_stream(const float* w1p, const float* w2f, const float* w0, int dim, int prec, std::vector<unsigned short>& out) {
  if ((dim != 128 && dim != 256) || (prec != PREC_BF16 && prec != PREC_F16)) return hipErrorInvalidValue;
  const int KT = dim / 64, NSS = dim / 32, NB128 = dim / 128;
  int perm[64];                          // stored k -> unit offset inside a 64-unit tile: bits 2 and 3 swapped per 16 units
  for (int k = 0; k < 64; ++k) perm[k] = (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1);
  out.clear();
  out.reserve((size_t)(NSS * (KT + NB128) + NB128 * KT) * FFN_PAIR / 2);
  if (w0)
    for (int kt = 0; kt < KT; ++kt)
      for (int rb = 0; rb < NB128; ++rb) append_tile(out, w0, dim, 128 * rb, 64 * kt, prec, nullptr);
  for (int ss = 0; ss < NSS; ++ss) {
    for (int kt = 0; kt < KT; ++kt)
      for (int half = 0; half < 2; ++half) append_tile(out, w1p, dim, 256 * ss + 128 * half, 64 * kt, prec, nullptr);
    for (int nb = 0; nb < NB128; ++nb)
      for (int kt2 = 0; kt2 < 2; ++kt2) append_tile(out, w2f, 5 * (size_t)dim, 128 * nb, 128 * ss + 64 * kt2, prec, perm);
  }
  for (int nb = 0; nb < NB128; ++nb)
    for (int kt = 0; kt < KT; ++kt) append_tile(out, w2f, 5 * (size_t)dim, 128 * nb, 4 * dim + 64 * kt, prec, nullptr);
  return hipSuccess;
}

bool ffn_eligible(int dim, int T, int prec) { return (dim == 128 || dim == 256) && T >= 64 && (prec == PREC_BF16 || prec == PREC_F16); }

template <typename TM, int D> static hipError_t launch_ffn_t(const FfnArgs& a, hipStream_t s) {
  if (a.att_q) {
    const size_t lds = FfnGeom<D, true>::LDS;
    hipLaunchKernelGGL((ffn_kernel<TM, D, true, true>), dim3(a.B * ((a.T + 63) / 64)), dim3(512), lds, s, a);
  } else if (a.pre_a) {
    const size_t lds = FfnGeom<D, true>::LDS;
    hipLaunchKernelGGL((ffn_kernel<TM, D, true>), dim3((a.M + 63) / 64), dim3(512), lds, s, a);
  } else {
    const size_t lds = FfnGeom<D, false>::LDS;
    hipLaunchKernelGGL((ffn_kernel<TM, D, false>), dim3((a.M + 63) / 64), dim3(512), lds, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_ffn(const FfnArgs& a, int prec, hipStream_t s) {
  if (!ffn_eligible(a.dim, a.T, prec) || a.M <= 0 || a.M != a.B * a.T) return hipErrorInvalidValue;
  if (!a.wstream || !a.consts || !a.bias2 || !a.res || (!a.out_f32 && !a.out_op)) return hipErrorInvalidValue;
  if (a.att_q) {      // in-kernel cross-attention: with the pre-stage's weights, bias and residual; 8 heads of dim / 8 channels, 16-byte aligned fragments
    if (!a.pre_bias || !a.pre_res || (a.pre_ldres & 3) || !a.att_kv || a.att_Lk < 1 || (a.att_ldq & 7) ||
        ((reinterpret_cast<uintptr_t>(a.att_q) | reinterpret_cast<uintptr_t>(a.att_kv)) & 15) || !(a.att_scale > 0.f))
      return hipErrorInvalidValue;
  } else if (a.pre_a ? (!a.pre_bias || !a.pre_res || (a.pre_lda & 7) || (a.pre_ldres & 3) || (unsigned long long)a.M * a.pre_lda * 2ull > 0xFFF00000ull)
              : (!a.yn || !a.ln_stats))
    return hipErrorInvalidValue;
  if ((!a.pre_a && !a.att_q && (a.ldy & 7)) || (a.ldres & 3) || (a.out_f32 && (a.ldo_f32 & 3)) || (a.out_op && (a.ldo_op & 3))) return hipErrorInvalidValue;
  if (!a.pre_a && !a.att_q && (unsigned long long)a.M * a.ldy * 2ull > 0xFFF00000ull) return hipErrorInvalidValue;
  if (prec == PREC_BF16) return a.dim == 128 ? launch_ffn_t<bf16_t, 128>(a, s) : launch_ffn_t<bf16_t, 256>(a, s);
  return a.dim == 128 ? launch_ffn_t<f16_t, 128>(a, s) : launch_ffn_t<f16_t, 256>(a, s);
}

// ---------------------------------------------------------------------------
// k | v fragment image for the in-kernel cross-attention (FfnArgs.att_kv; once per utterance, beside the hoisted k | v projection)
// ---------------------------------------------------------------------------
template <typename TM>
__global__ __launch_bounds__(256) void xattn_pack_kernel(const TM* __restrict__ k, int ldk, const TM* __restrict__ v, int ldv, int Lk, int ntile, int hd, uint4* __restrict__ out) {
  // one thread per 16-byte piece: piece = ((bh * ntile + t) * (nsl + 2) + frag) * 64 + lane
  const int nsl = hd >> 4, nfr = nsl + 2;
  const size_t piece = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)gridDim.y * ntile * nfr * 64;         // gridDim.y = B * 8 heads (pieces of one (b, h) never straddle: see the launcher)
  (void)total;
  const int bh = blockIdx.y;
  if (piece >= (size_t)ntile * nfr * 64) return;
  const int lane = (int)(piece & 63), frag = (int)((piece >> 6) % nfr), t = (int)((piece >> 6) / nfr);
  const int b = bh >> 3, hh = bh & 7, l31 = lane & 31, hi = lane >> 5;
  union { uint4 u; uint16_t e[8]; } r;
  r.u = make_uint4(0u, 0u, 0u, 0u);
  const uint16_t one = std::is_same<TM, f16_t>::value ? (uint16_t)0x3C00 : (uint16_t)0x3F80;
  if (frag < nsl) {                      // K fragment: key = 32 t + l31, channels 16 frag + 8 hi ..
    const int key = 32 * t + l31;
    if (key < Lk) r.u = *reinterpret_cast<const uint4*>(k + (size_t)(b * Lk + key) * ldk + hh * hd + 16 * frag + 8 * hi);
  } else {                               // V^T fragment j: row d = l31, slots 16 j + 8 hi + e of the tile; key bits 2 and 3 swapped inside a group of 16
    const int j = frag - nsl, d = l31;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int slot = 16 * j + 8 * hi + e;
      const int key = 32 * t + ((slot & ~12) | ((slot & 4) << 1) | ((slot & 8) >> 1));
      if (d < hd) { if (key < Lk) r.e[e] = v[(size_t)(b * Lk + key) * ldv + hh * hd + d].v; }
      else if (d == hd && hd == 16) r.e[e] = one;
    }
  }
  out[((size_t)bh * ntile * nfr) * 64 + piece] = r.u;
}
size_t xattn_pack_bytes(int B, int Lk, int hd) { return (size_t)B * 8 * (size_t)((Lk + 31) >> 5) * (size_t)((hd >> 4) + 2) * 1024; }
hipError_t launch_xattn_pack(const void* k, int ldk, const void* v, int ldv, int B, int Lk, int hd, void* out, int prec, hipStream_t s) {
  if (!k || !v || !out || B < 1 || Lk < 1 || (hd != 16 && hd != 32) || (prec != PREC_BF16 && prec != PREC_F16) || (ldk & 7) ||
      (reinterpret_cast<uintptr_t>(k) & 15))
    return hipErrorInvalidValue;
  const int ntile = (Lk + 31) >> 5, nfr = (hd >> 4) + 2;
  dim3 grid((ntile * nfr * 64 + 255) / 256, B * 8);
  if (prec == PREC_BF16) hipLaunchKernelGGL(xattn_pack_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, Lk, ntile, hd, (uint4*)out);
  else hipLaunchKernelGGL(xattn_pack_kernel<f16_t>, grid, dim3(256), 0, s, (const f16_t*)k, ldk, (const f16_t*)v, ldv, Lk, ntile, hd, (uint4*)out);
  return hipGetLastError();
}

hipError_t init_ffn_attributes() {
  hipError_t e;
#define NS2VC_FFN_ATTR(TM, D_, PRE_)                                                                                          \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_kernel<TM, D_, PRE_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)FfnGeom<D_, PRE_>::LDS)) != hipSuccess) return e
  NS2VC_FFN_ATTR(bf16_t, 128, false); NS2VC_FFN_ATTR(bf16_t, 256, false); NS2VC_FFN_ATTR(f16_t, 128, false); NS2VC_FFN_ATTR(f16_t, 256, false);
  NS2VC_FFN_ATTR(bf16_t, 128, true); NS2VC_FFN_ATTR(bf16_t, 256, true); NS2VC_FFN_ATTR(f16_t, 128, true); NS2VC_FFN_ATTR(f16_t, 256, true);
#undef NS2VC_FFN_ATTR
#define NS2VC_FFN_ATTR2(TM, D_)                                                                                                      \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_kernel<TM, D_, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)FfnGeom<D_, true>::LDS)) != hipSuccess) return e
  NS2VC_FFN_ATTR2(bf16_t, 128); NS2VC_FFN_ATTR2(bf16_t, 256); NS2VC_FFN_ATTR2(f16_t, 128); NS2VC_FFN_ATTR2(f16_t, 256);
#undef NS2VC_FFN_ATTR2
  return hipSuccess;
}

}  // namespace ns2vc
