// Two token-local GEMMs of a transformer block in ONE launch, CDNA4 (gfx950), 16-bit operand types.
//
// Inside BasicTransformerBlock (reference unet1d/attention.py:130-176) and Transformer2DModel (transformer_1d.py:256-286) two
// pairs of linears follow each other with nothing but a LayerNorm in between, and both act on every token by itself:
//
//   proj_in (1x1 conv) -> norm1 -> to_q | to_k | to_v          (transformer_1d.py:270-279, attention.py:130-140)
//   attn1.to_out + residual -> norm2 -> attn2.to_q               (attention_processor.py:1040-1050, attention.py:141-160)
//
//     y   = A W1^T + b1 (+ res)                     -> fp32 residual stream (kept), LayerNorm statistics of the row
//     z   = LayerNorm(y) W2^T + b2                  =  rstd (y_op W2'^T - mean rowsum(W2')) + b2'     (gamma/beta folded at pack time)
//
// As separate launches every one of these small-K GEMMs lasts one workgroup latency (8-19 us at 10 s x batch 32: set-up, a
// cold first tile, an LDS-staged epilogue, a store drain) and y's operand copy makes a round trip through HBM.  Here one
// workgroup owns 64 tokens for the chain, exactly like the fused feed-forward (ffn.hip): the token panel sits in LDS, the
// weights are ONE flat stream of pre-swizzled 16-KB tiles consumed in pairs through a ring by lane-linear LDS-DMA, and
// everything is computed TRANSPOSED (out^T = W a^T) so that a lane owns ONE token: bias, residual, the LayerNorm sums and
// the fix-up are per-lane arithmetic, y's operand copy is written straight back into the panel (it replaces A, which is
// dead by then) and never leaves the CU.  8 waves = 2 token halves x 4 row groups: wave (tw, cg) takes rows 32 cg .. + 31
// of EVERY 128-row weight tile for its 32 tokens.
#include "common.h"
#include "mma.h"
#include <cstdlib>
#include <vector>

namespace ns2vc {

typedef ::ns2vc_rowchain_args RowchainArgs;

// optional per-workgroup phase timing (cycles, thread 0): [block][8] = entry, prologue end (panel / constants / first pairs issued, GroupNorm
// applied), stage-1 loop end, stage-1 epilogue end, stage-2 loop end, exit (stores drained).  `make TRACE=1` builds only;
// set through ns2vc_debug_set_gemm_trace, read by tools/rowchain_trace.py
__device__ unsigned long long* g_rc_trace = nullptr;
#ifndef NS2VC_GEMM_TRACE
#define NS2VC_GEMM_TRACE 0
#endif
#ifndef NS2VC_RC_ABLATE
#define NS2VC_RC_ABLATE 0
#endif
#ifndef NS2VC_RC_WT
#define NS2VC_RC_WT 1       // stage-2 result stores write-through (sc1), as the GEMM epilogues' (common.h out_store16)
#endif
#if NS2VC_GEMM_TRACE
#define RC_TR(i) do { if (tr && tid == 0) tr[i] = __builtin_readcyclecounter(); } while (0)
#else
#define RC_TR(i) do { (void)tr; } while (0)
#endif
void set_rc_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rc_trace), &p, sizeof(p)); }

constexpr int RC_TILE = 128 * 128;        // bytes: [128 rows][128 B of K]
constexpr int RC_PAIR = 2 * RC_TILE;

template <int D, int R2, int NT> struct RowchainGeom {
  static constexpr int TOK = 64 * NT;       // tokens per workgroup: every wave owns NT blocks of 32
  static constexpr int KT = D / 64;         // K tiles of both stages (K = D)
  static constexpr int NB1 = D / 128;       // 128-row blocks of W1 (N1 = D)
  static constexpr int S1 = KT * NB1, S2 = KT * R2;           // tiles per stage, k-tile outer, row block inner
  static_assert(S1 % 2 == 0 && S2 % 2 == 0, "stages start on pair boundaries");
  static constexpr int NP = (S1 + S2) / 2;
  static constexpr int RING = 3;
  static constexpr int PTILE = TOK * 128;                     // bytes of one panel k tile: [TOK][128 B], swizzled
  static constexpr int PANEL = KT * PTILE;
  static constexpr int CONSTS = R2 * 128 * 8;                 // (rowsum, bias) per stage-2 row
  static constexpr int STATS = TOK * 4 * 8;                   // (sum, sumsq) per token and row group
  static constexpr int LDS = RING * RC_PAIR + PANEL + CONSTS + STATS;
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <typename TM, int D, int R2, int NT>
__global__ __launch_bounds__(512) void rowchain_kernel(const RowchainArgs a) {
  op_mode_init<TM>();
  using G = RowchainGeom<D, R2, NT>;
  constexpr int KT = G::KT, NB1 = G::NB1, S1 = G::S1, S2 = G::S2, NP = G::NP, RING = G::RING, TOK = G::TOK, PTILE = G::PTILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  char* const panel = smem + RING * RC_PAIR;
  const float* const consts = reinterpret_cast<const float*>(panel + G::PANEL);
  float2* const stats = reinterpret_cast<float2*>(panel + G::PANEL + G::CONSTS);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long* const tr = (NS2VC_GEMM_TRACE && g_rc_trace) ? g_rc_trace + (size_t)blockIdx.x * 8 : nullptr;
  RC_TR(0);
  const int tw = wave & 1, cg = wave >> 1;          // token half, row group
  const int l31 = lane & 31, hi = lane >> 5;
  const int sw = (l31 >> 1) & 7;                    // XOR swizzle of every fragment / panel row this lane touches
  const unsigned lds0 = (unsigned)(size_t)smem;
  // N-sliced stage 2 (r4, dim 384 where 64-token blocks fill less than half of the chip): `S` workgroups per token block, each repeating
  // stage 1 and taking R2 of the stage-2 row blocks.  Workgroup ids are handed out round-robin over the 8 XCDs, so the S slices of a block
  // sit 8 ids apart: they run on the same XCD and share the fp32 rows / panel through its L2.
  const int S = a.slices > 1 ? a.slices : 1;
  const int sl = S > 1 ? (int)(blockIdx.x % (8 * S)) / 8 : 0;
  const int blk = S > 1 ? (int)(blockIdx.x / (8 * S)) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
  const int m0 = blk * TOK;
  if (m0 >= a.M) return;                            // (padding of the sliced grid to whole groups of 8; uniform over the workgroup)
  const int tok0 = 32 * NT * tw + l31;              // this lane's tokens inside the block: tok0 + 32 u (both lane halves)

  const i32x4_t rW = make_rsrc(reinterpret_cast<const char*>(a.wstream) + (size_t)sl * NP * RC_PAIR, (unsigned long long)NP * RC_PAIR);
  const unsigned lane16 = (unsigned)(lane * 16);
  // ---- token panel.  Plain case: LDS-DMA of the operand rows (source-side swizzle, rows past M read as zeros).
  // GroupNorm case (a.gn_x: the panel is GroupNorm(x), the `norm` of Transformer2DModel, transformer_1d.py:268): the fp32 rows
  // of x are normalised here with the statistics the producing GEMM's epilogue left -- the same arithmetic as
  // gn_apply_kernel (misc.hip), so the panel is bit-identical to what that launch would have written.
  if (a.gn_x == nullptr) {
    const int pchunk = lane & 7;
    const i32x4_t rA = make_rsrc(a.a_op, (unsigned long long)a.M * a.lda * 2ull);
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int prow = 64 * u + 8 * wave + (lane >> 3);                  // one 1-KB piece per wave = 8 rows x 128 B
      const int m = m0 + prow;
      const unsigned voff = m < a.M ? (unsigned)m * (unsigned)a.lda * 2u + (unsigned)((pchunk ^ ((prow >> 1) & 7)) * 16) : DMA_OOB;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
        blds16(rA, voff, (unsigned)(kt * 128), lds0 + RING * RC_PAIR + kt * PTILE + u * 8192 + wave * 1024);
    }
  }
  {
    // (rowsum, bias) of this slice's stage-2 rows; rows past n2 (the short last slice) lie outside the descriptor and arrive as zeros
    const i32x4_t rC = make_rsrc(a.consts2, (unsigned long long)a.n2 * 8ull);
#pragma unroll
    for (int pc = 0; pc < (G::CONSTS / 1024 + 7) / 8; ++pc)
      if (8 * pc + wave < G::CONSTS / 1024)
        blds16(rC, lane16 + (unsigned)(sl * G::CONSTS), (unsigned)((8 * pc + wave) * 1024), lds0 + RING * RC_PAIR + G::PANEL + (8 * pc + wave) * 1024);
  }
  static_assert(G::CONSTS % 1024 == 0, "constants are whole DMA pieces");
  auto issue_pair = [&](int p) __attribute__((always_inline)) {
    const int slot = p % RING;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      blds16(rW, lane16, (unsigned)(p * RC_PAIR + (j * 8 + wave) * 1024), lds0 + slot * RC_PAIR + (j * 8 + wave) * 1024);
  };
#pragma unroll
  for (int q = 0; q < RING - 1; ++q)
    if (q < NP) issue_pair(q);

  // stage-1 bias of this lane's rows: row = 128 rb + 32 cg + 8 g + 4 hi + i  <->  register 4 g + i of block rb
  float4 b1[NB1][4];
#pragma unroll
  for (int rb = 0; rb < NB1; ++rb)
#pragma unroll
    for (int g = 0; g < 4; ++g) b1[rb][g] = *reinterpret_cast<const float4*>(a.bias1 + 128 * rb + 32 * cg + 8 * g + 4 * hi);
  // residual rows of this lane's tokens: independent of everything else, so their latency hides under the first tiles
  // (res may alias out1 element for element: the same lane reads here and writes in the stage-1 epilogue)
  float4 rr[NT][NB1][4];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int rb = 0; rb < NB1; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m = m0 + tok0 + 32 * u;
        rr[u][rb][g] = (a.res && m < a.M) ? *reinterpret_cast<const float4*>(a.res + (size_t)m * a.ldres + 128 * rb + 32 * cg + 8 * g + 4 * hi)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }

  if (a.gn_x) {
    // float4 quads per row, rows per pass (dim 384: 96 quads, 5 rows = 480 of the 512 threads), passes over the TOK rows
    constexpr int QPR = D / 4, RPP = 512 / QPR, NPASS = (TOK + RPP - 1) / RPP;
    const bool act = tid < RPP * QPR;
    const int quad = act ? tid % QPR : 0, r0 = act ? tid / QPR : 0, c = 4 * quad;
    float4 xv[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + r0, m = m0 + row;
      xv[ps] = (m < a.M && row < TOK) ? *reinterpret_cast<const float4*>(a.gn_x + (size_t)m * a.ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 ga = *reinterpret_cast<const float4*>(a.gn_gamma + c), be = *reinterpret_cast<const float4*>(a.gn_beta + c);
    // (mean, rstd) of every (batch item touched by this block, group): at most 4 items (the launcher checks T)
    const int Cg = D / a.G, nb = Cg >> 4, nblk = D >> 4;
    const int b_lo = m0 / a.T;
    const int nbi = (min(m0 + TOK, a.M) - 1) / a.T - b_lo + 1;
    float2* const gtab = stats;                      // free until the stage-1 epilogue
    if (tid < nbi * a.G) {
      const int bi = tid / a.G, g = tid - bi * a.G;
      const long long* st = a.gn_stats + ((size_t)(b_lo + bi) * nblk + (size_t)g * nb) * 2;
      double ds = 0.0, dq = 0.0;
      for (int j = 0; j < nb; ++j) { ds += (double)st[2 * j] * (1.0 / GN_SUM_SCALE); dq += (double)st[2 * j + 1] * (1.0 / GN_SQ_SCALE); }
      const float inv_nf = 1.0f / ((float)a.T * (float)Cg);
      const double inv_n = (double)inv_nf * (2.0 - (double)inv_nf * ((double)a.T * (double)Cg));
      const double mean = ds * inv_n;
      double var = dq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float ve = (float)var + a.gn_eps;
      float r = rsqrtf(ve);
      r = r * (1.5f - 0.5f * ve * r * r);
      gtab[bi * 8 + g] = make_float2((float)mean, r);
    }
    __syncthreads();
    const int g = c / Cg;
    float2 mrs[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int m = m0 + ps * RPP + r0;
      int bi = 0;                                    // batch item of this row, relative to b_lo (no division)
#pragma unroll
      for (int k = 1; k < 4; ++k) bi += (m >= (b_lo + k) * a.T) ? 1 : 0;
      mrs[ps] = gtab[min(bi, 3) * 8 + g];
    }
    // gfx950 hazard guard (profiles/r04_gn_prologue_rootcause.txt): every (mean, rstd) pair has landed before the first packed
    // fp32 product is formed from them (the compiler's counted lgkmcnt(N) waits in front of v_pk_* with op_sel is the pattern
    // that returned zeros in gemm.hip's prologue; tools/isa_pk_lds_check.py keeps the library free of it)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) asm volatile("" : "+v"(mrs[ps].x), "+v"(mrs[ps].y));
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + r0, m = m0 + row;
      const float2 mr = mrs[ps];
      const float s0 = mr.y * ga.x, s1 = mr.y * ga.y, s2 = mr.y * ga.z, s3 = mr.y * ga.w;
      float y0 = xv[ps].x * s0 + (be.x - mr.x * s0), y1 = xv[ps].y * s1 + (be.y - mr.x * s1);
      float y2 = xv[ps].z * s2 + (be.z - mr.x * s2), y3 = xv[ps].w * s3 + (be.w - mr.x * s3);
      if (m >= a.M) { y0 = 0.f; y1 = 0.f; y2 = 0.f; y3 = 0.f; }
      char* dst = panel + (c >> 6) * PTILE + row * 128 + ((((c & 63) >> 3) ^ ((row >> 1) & 7)) * 16) + (c & 4) * 2;
      if (act && row < TOK) *reinterpret_cast<uint2*>(dst) = make_uint2(Op16<TM>::pack(y0, y1), Op16<TM>::pack(y2, y3));
    }
    // (no barrier here: the first step_begin drains the LDS writes and synchronises the block before any fragment read)
  }

  RC_TR(1);
  int p = 0;                                        // next pair to consume (compile-time after unrolling)
  auto step_begin = [&](bool first = false) __attribute__((always_inline)) -> const char* {
    // pair p has landed when only the pieces (four per wave and pair) of the pair behind it may still be in flight.  Loads
    // complete in issue order among themselves, so other outstanding memory operations (the result stores of the stage-1
    // epilogue) can only make this wait longer than necessary, never shorter.
    if (NP - 1 - p >= 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads of the slot about to be refilled are done
    __builtin_amdgcn_s_barrier();
    if (first) {
      // The compiler does not see the DMA instructions.  Left alone it waits for the bias / residual rows (issued above)
      // only where the stage-1 epilogue first uses them -- with `s_waitcnt vmcnt(0)`, which by then also covers the
      // prefetched stage-2 weight pairs and, from the second register group on, every result STORE issued so far: eight
      // serialised store round trips per wave.  Consuming the registers here, before any younger memory operation exists,
      // moves that wait to a point where it costs nothing.
#pragma unroll
      for (int rb = 0; rb < NB1; ++rb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          asm volatile("" : "+v"(b1[rb][g].x), "+v"(b1[rb][g].y), "+v"(b1[rb][g].z), "+v"(b1[rb][g].w));
#pragma unroll
          for (int u = 0; u < NT; ++u) asm volatile("" : "+v"(rr[u][rb][g].x), "+v"(rr[u][rb][g].y), "+v"(rr[u][rb][g].z), "+v"(rr[u][rb][g].w));
        }
    }
    if (p + RING - 1 < NP) issue_pair(p + RING - 1);
    return ring + (p % RING) * RC_PAIR;
  };
  const char* const bpanel = panel + tok0 * 128;    // this lane's first token row inside a panel tile (+ kt * PTILE + u * 4096)

  // one 16-KB tile: acc[u] += W[rows 32 cg ..][64 k] * panel[kt][tokens of block u]^T; the weight fragments serve all NT blocks
  auto tile_mma = [&](f32x16_t (&acc)[NT], const char* T, int kt) __attribute__((always_inline)) {
    const char* wrow = T + (32 * cg + l31) * 128;
    u32x4_t fb[NT][4], fw[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ((2 * ks + hi) ^ sw) * 16;
      fw[ks] = *reinterpret_cast<const u32x4_t*>(wrow + c);
#pragma unroll
      for (int u = 0; u < NT; ++u) fb[u][ks] = *reinterpret_cast<const u32x4_t*>(bpanel + kt * PTILE + u * 4096 + c);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int u = 0; u < NT; ++u) MmaT<TM>::mma(acc[u], fw[ks], fb[u][ks]);
  };

  // ---- stage 1: y^T = W1 A^T
  f32x16_t acc1[NB1][NT];
#pragma unroll
  for (int i = 0; i < NB1; ++i)
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][u][r] = 0.f;
#pragma unroll
  for (int j = 0; j < S1 / 2; ++j) {
    const char* T = step_begin(j == 0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = 2 * j + t;
      tile_mma(acc1[i % NB1], T + t * RC_TILE, i / NB1);
    }
    ++p;
  }

  // ---- stage-1 epilogue, per lane: + bias (+ fp32 residual), fp32 result row segments, LayerNorm sums, operand copy
  // into the panel
  RC_TR(2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                     // every wave is done reading A: the panel may be overwritten with y
  float mean[NT], rstd[NT];
  // fp32 rows of y: a lane owns ONE token, so a direct store writes 32 B to each of 32 different rows per instruction --
  // measured (tools/rowchain_trace.py) at 5-7 k cycles of this epilogue.  Instead every wave turns its 32 token x 32 channel
  // block around through 4 KB of the ring slot that is free between the stages (wave-private, XOR-swizzled 16-B chunks, no
  // barrier), so that 8 lanes write the 128 contiguous bytes of one row.
  char* const wscr = ring + ((p + RING - 1) % RING) * RC_PAIR + wave * 4096;     // slot of the last stage-1 pair
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int tok = tok0 + 32 * u, mtok = m0 + tok;
    const bool tok_ok = mtok < a.M;
    float ps = 0.f, pq = 0.f;
#pragma unroll
    for (int rb = 0; rb < NB1; ++rb) {
      float4 vq[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v;
        v.x = acc1[rb][u][4 * g + 0] + b1[rb][g].x + rr[u][rb][g].x;
        v.y = acc1[rb][u][4 * g + 1] + b1[rb][g].y + rr[u][rb][g].y;
        v.z = acc1[rb][u][4 * g + 2] + b1[rb][g].z + rr[u][rb][g].z;
        v.w = acc1[rb][u][4 * g + 3] + b1[rb][g].w + rr[u][rb][g].w;
        const int n = 128 * rb + 32 * cg + 8 * g + 4 * hi;
        if (tok_ok) {
          ps += (v.x + v.y) + (v.z + v.w);
          pq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        } else {
          v = make_float4(0.f, 0.f, 0.f, 0.f);      // rows past M: a zero operand row (nothing of it is stored)
        }
        vq[g] = v;
        // operand copy: channel n -> k tile n / 64, 16-B chunk (n % 64) / 8, bytes 8 hi .. + 7 of the chunk
        char* dst = panel + (n >> 6) * PTILE + tok * 128 + ((((n & 63) >> 3) ^ sw) * 16) + 8 * hi;
        *reinterpret_cast<uint2*>(dst) = make_uint2(Op16<TM>::pack(v.x, v.y), Op16<TM>::pack(v.z, v.w));
      }
      if (a.out1_f32 && sl == 0) {                  // (every slice computes the same y: the first one stores it)
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(wscr + l31 * 128 + (((2 * g + hi) ^ (l31 & 7)) * 16)) = vq[g];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 wv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int T = 8 * k + (lane >> 3), seg = lane & 7;
          wv[k] = *reinterpret_cast<const float4*>(wscr + T * 128 + ((seg ^ (T & 7)) * 16));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int T = 8 * k + (lane >> 3), seg = lane & 7;
          const int mt = m0 + 32 * NT * tw + 32 * u + T;
          if (mt < a.M) out_f4(a.out1_f32 + (size_t)mt * a.ldo1 + 128 * rb + 32 * cg + 4 * seg, wv[k].x, wv[k].y, wv[k].z, wv[k].w);
        }
      }
    }
    // the two lane halves of a token, then the four row groups through LDS
    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ps), __float_as_uint(ps), false, false);
    const auto q2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(pq), __float_as_uint(pq), false, false);
    ps = __uint_as_float(s2[0]) + __uint_as_float(s2[1]);
    pq = __uint_as_float(q2[0]) + __uint_as_float(q2[1]);
    if (hi == 0) stats[tok * 4 + cg] = make_float2(ps, pq);
  }
  lds_barrier();                                    // y panel and statistics complete (LDS-only: the fp32 stores of y stay in flight)
  {
    float ratio = 0.f;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int tok = tok0 + 32 * u;
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float2 v = stats[tok * 4 + k]; s += v.x; q += v.y; }   // fixed order: deterministic
      const float inv = 1.0f / (float)D;
      mean[u] = s * inv;
      double var = (double)q * (double)inv - (double)mean[u] * (double)mean[u];
      if (var < 0.0) var = 0.0;
      rstd[u] = 1.0f / sqrtf((float)var + a.ln_eps);
      if (m0 + tok < a.M) ratio = fmaxf(ratio, fabsf(mean[u]) * rstd[u]);
    }
    if (a.ln_health && cg == 0 && sl == 0) {          // same health report as the LayerNorm-consumer GEMMs (gemm.hip ln_row_finish)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
      if (lane == 0 && ratio > __uint_as_float(__hip_atomic_load(a.ln_health, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(a.ln_health, __float_as_uint(ratio));
    }
  }

  // ---- stage 2: z^T = W2' y_op^T, all R2 row blocks accumulate at once (k-tile outer)
  RC_TR(3);
  f32x16_t acc2[R2][NT];
#pragma unroll
  for (int i = 0; i < R2; ++i)
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][u][r] = 0.f;
#pragma unroll
  for (int j = 0; j < S2 / 2; ++j) {
    const char* T = step_begin();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = 2 * j + t;
      tile_mma(acc2[i % R2], T + t * RC_TILE, i / R2);
    }
    ++p;
  }

  RC_TR(4);
  // ---- stage-2 epilogue: LayerNorm fix-up + bias per element, operand rows out.  A lane holds 4 consecutive channels per register group g of
  // ONE token: stored from the registers, every instruction is 32 rows x 32 bytes, and the issue of those pieces was the tail of the kernel (r5:
  // without the stage-2 stores the step ran 3 % faster).  The weights and the panel are dead by now, so the whole result tile goes through their
  // LDS ([token][RBP x 128 channels], row pitch + 16 B: the eight tokens of a write group land in different banks) and leaves as contiguous
  // 1-KB runs, 64 lanes x 16 B.  RBP row blocks per pass (all of them where they fit).
  TM* const oo = reinterpret_cast<TM*>(a.out2_op);
  constexpr int AVAIL = RING * RC_PAIR + G::PANEL;
  constexpr int RBP = TOK * (R2 * 256 + 16) <= AVAIL ? R2 : (R2 + 1) / 2;
  static_assert(TOK * (RBP * 256 + 16) <= AVAIL, "a pass of the staged result tile fits in the ring + panel");
  constexpr int PB = RBP * 256 + 16;                // staged row pitch (bytes)
  constexpr int CPR = RBP * 16;                     // 16-byte chunks per staged row
#pragma unroll
  for (int rb0 = 0; rb0 < R2; rb0 += RBP) {
    lds_barrier();                                  // every wave is done with the ring / the panel (first pass), with reading the previous pass
#pragma unroll
    for (int rbi = 0; rbi < RBP; ++rbi) {
      const int rb = rb0 + rbi;
      if (rb < R2) {
        float4 c0[4], c1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* cv = consts + (size_t)(128 * rb + 32 * cg + 8 * g + 4 * hi) * 2;      // (rowsum, bias) x 4 rows
          c0[g] = *reinterpret_cast<const float4*>(cv); c1[g] = *reinterpret_cast<const float4*>(cv + 4);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (hazard guard, see the GroupNorm prologue above: no packed fp32 op under outstanding LDS reads here)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(c0[g].x), "+v"(c0[g].y), "+v"(c0[g].z), "+v"(c0[g].w), "+v"(c1[g].x), "+v"(c1[g].y), "+v"(c1[g].z), "+v"(c1[g].w));
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          uint32_t pk[4][2];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float z0 = rstd[u] * (acc2[rb][u][4 * g + 0] - mean[u] * c0[g].x) + c0[g].y;
            const float z1 = rstd[u] * (acc2[rb][u][4 * g + 1] - mean[u] * c0[g].z) + c0[g].w;
            const float z2 = rstd[u] * (acc2[rb][u][4 * g + 2] - mean[u] * c1[g].x) + c1[g].y;
            const float z3 = rstd[u] * (acc2[rb][u][4 * g + 3] - mean[u] * c1[g].z) + c1[g].w;
            pk[g][0] = Op16<TM>::pack(z0, z1);
            pk[g][1] = Op16<TM>::pack(z2, z3);
          }
          char* srow = smem + (size_t)(tok0 + 32 * u) * PB + rbi * 256 + 64 * cg + 16 * hi;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {         // groups (2 gp, 2 gp + 1): lower half ends up with group 2 gp, upper half with 2 gp + 1 -- 8 channels, 16 B
            const auto x0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
            const auto x1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
            *reinterpret_cast<u32x4_t*>(srow + 32 * gp) = u32x4_t{x0[0], x1[0], x0[1], x1[1]};
          }
        }
      }
    }
    lds_barrier();                                  // the pass is staged
    const int nb = min(RBP, R2 - rb0);              // row blocks really staged (the last pass may be short)
#pragma unroll
    for (int i = 0; i < TOK * CPR / 512; ++i) {
      const int id = i * 512 + tid;
      const int row = id / CPR, col = id - row * CPR;
      const int mt = m0 + row, n = 128 * (rb0 + sl * R2) + 8 * col;
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + (size_t)row * PB + col * 16);
#if NS2VC_RC_ABLATE & 1     // diagnostic build (wrong results, timing only): no stage-2 stores
      asm volatile("" :: "v"(v));
#else
#if NS2VC_RC_WT
      if (mt < a.M && col < nb * 16 && n < a.n2) out_store16(oo + (size_t)mt * a.ldo2 + n, v.x, v.y, v.z, v.w);
#else
      if (mt < a.M && col < nb * 16 && n < a.n2) *reinterpret_cast<u32x4_t*>(oo + (size_t)mt * a.ldo2 + n) = v;
#endif
#endif
    }
  }
#if NS2VC_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (only so that the last stamp includes the store drain)
#endif
  RC_TR(5);
}

// ---------------------------------------------------------------------------
// host side: tile-stream packer and launcher
// ---------------------------------------------------------------------------
// one 16 KB tile [128 rows][64 k] of a row-major fp32 matrix, rounded to the operand type, in the swizzled LDS image:
// byte r*128 + pos*16 holds logical chunk pos ^ ((r>>1)&7) of row r
static void rc_append_tile(std::vector<unsigned short>& out, const float* mat, size_t ld, int row0, int col0, int prec) {
  for (int r = 0; r < 128; ++r)
    for (int pos = 0; pos < 8; ++pos) {
      const int lc = pos ^ ((r >> 1) & 7);
      for (int e = 0; e < 8; ++e) out.push_back(f32_to_op16_bits(mat[(size_t)(row0 + r) * ld + col0 + lc * 8 + e], prec));
    }
}

// w1 [dim][dim], w2 [n2][dim] (LayerNorm-folded), both fp32 host row-major (K contiguous)
// stage-2 row blocks per slice of the N-sliced launch (r4): q|k|v at dim 384 as 5 + 4, to_q as 2 + 1
int rowchain_slice_blocks(int n2, int slices) { return (n2 / 128 + slices - 1) / slices; }

hipError_t pack_rowchain_stream(const float* w1, const float* w2, int dim, int n2, int prec, std::vector<unsigned short>& out, int slices) {
  if (!rowchain_eligible(dim, n2, 64, prec) || slices < 1 || (slices > 1 && dim != 384)) return hipErrorInvalidValue;
  const int KT = dim / 64, nb2 = n2 / 128, r2s = rowchain_slice_blocks(n2, slices);
  out.clear();
  out.reserve((size_t)slices * (dim + r2s * 128) * dim);
  std::vector<float> zero((size_t)128 * dim, 0.f);
  for (int sl = 0; sl < slices; ++sl) {            // every slice: the whole of stage 1, then its own stage-2 rows (a missing block = zero tiles)
    for (int kt = 0; kt < KT; ++kt)
      for (int rb = 0; rb < dim / 128; ++rb) rc_append_tile(out, w1, dim, 128 * rb, 64 * kt, prec);
    for (int kt = 0; kt < KT; ++kt)
      for (int j = 0; j < r2s; ++j) {
        const int rb = sl * r2s + j;
        if (rb < nb2) rc_append_tile(out, w2, dim, 128 * rb, 64 * kt, prec);
        else rc_append_tile(out, zero.data(), dim, 0, 64 * kt, prec);
      }
  }
  return hipSuccess;
}

bool rowchain_eligible(int dim, int n2, int T, int prec) {
  return (dim == 128 || dim == 256 || dim == 384) && (n2 == dim || n2 == 3 * dim) && T >= 1 && (prec == PREC_BF16 || prec == PREC_F16);
}

template <typename TM, int D, int R2, int NT> static hipError_t launch_rc(const RowchainArgs& a, hipStream_t s) {
  const size_t lds = RowchainGeom<D, R2, NT>::LDS;
  constexpr int TOK = RowchainGeom<D, R2, NT>::TOK;
  const int nblk = (a.M + TOK - 1) / TOK, S = a.slices > 1 ? a.slices : 1;
  hipLaunchKernelGGL((rowchain_kernel<TM, D, R2, NT>), dim3(S > 1 ? ((nblk + 7) / 8) * 8 * S : nblk), dim3(512), lds, s, a);
  return hipGetLastError();
}
// 64-token workgroups by default.  At dim 128 a workgroup's weights are small and M is large: when 64-token blocks would not
// fit the chip in one round (one workgroup per CU: the LDS ring), 128-token blocks halve the grid and the weight traffic
static int g_force_nt = 0;   // test / tuning hook (ns2vc_debug_set_rowchain_tokens): 1 / 2 forces the block size (2 only exists for dim 128)
void set_forced_rowchain_tokens(int nt) { g_force_nt = nt; }
template <typename TM> static hipError_t launch_rc_tm(const RowchainArgs& a, hipStream_t s) {
  if (a.dim == 128) {
    const bool big = g_force_nt ? g_force_nt == 2 : (a.M + 63) / 64 > 256;
    if (big) return a.n2 == 128 ? launch_rc<TM, 128, 1, 2>(a, s) : launch_rc<TM, 128, 3, 2>(a, s);
    return a.n2 == 128 ? launch_rc<TM, 128, 1, 1>(a, s) : launch_rc<TM, 128, 3, 1>(a, s);
  }
  if (a.dim == 384 && a.slices == 2) return a.n2 == 384 ? launch_rc<TM, 384, 2, 1>(a, s) : launch_rc<TM, 384, 5, 1>(a, s);
  if (a.dim == 384) return a.n2 == 384 ? launch_rc<TM, 384, 3, 1>(a, s) : launch_rc<TM, 384, 9, 1>(a, s);
  return a.n2 == 256 ? launch_rc<TM, 256, 2, 1>(a, s) : launch_rc<TM, 256, 6, 1>(a, s);
}

hipError_t launch_rowchain(const RowchainArgs& a, int prec, hipStream_t s) {
  if (!rowchain_eligible(a.dim, a.n2, 64, prec) || a.M <= 0) return hipErrorInvalidValue;
  if (a.slices > 1 && (a.slices != 2 || a.dim != 384 || (a.res && (const void*)a.res == (const void*)a.out1_f32))) return hipErrorInvalidValue;   // (in-place residual: the slices would race on y)
  if ((!a.a_op && !a.gn_x) || !a.wstream || !a.bias1 || !a.consts2 || !a.out2_op) return hipErrorInvalidValue;
  if (a.gn_x) {      // GroupNorm prologue: per-16-channel-block statistics, groups of whole blocks, <= 4 batch items per 128 tokens
    if (!a.gn_stats || !a.gn_gamma || !a.gn_beta || a.G < 1 || a.G > 8 || a.T < 64 || (a.dim % a.G) || ((a.dim / a.G) & 15) || (a.ldx & 3))
      return hipErrorInvalidValue;
  } else if (a.lda & 7) return hipErrorInvalidValue;
  if ( (a.res && (a.ldres & 3)) || (a.out1_f32 && (a.ldo1 & 3)) || (a.ldo2 & 7)) return hipErrorInvalidValue;
  if ((unsigned long long)a.M * a.lda * 2ull > 0xFFF00000ull) return hipErrorInvalidValue;
  return prec == PREC_BF16 ? launch_rc_tm<bf16_t>(a, s) : launch_rc_tm<f16_t>(a, s);
}

hipError_t init_rowchain_attributes() {
  hipError_t e;
#define NS2VC_RC_ATTR(TM, DD_, RR_, NT_)                                                                                            \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(rowchain_kernel<TM, DD_, RR_, NT_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)RowchainGeom<DD_, RR_, NT_>::LDS)) != hipSuccess) return e
  NS2VC_RC_ATTR(bf16_t, 128, 1, 1); NS2VC_RC_ATTR(bf16_t, 128, 3, 1); NS2VC_RC_ATTR(bf16_t, 256, 2, 1); NS2VC_RC_ATTR(bf16_t, 256, 6, 1);
  NS2VC_RC_ATTR(f16_t, 128, 1, 1); NS2VC_RC_ATTR(f16_t, 128, 3, 1); NS2VC_RC_ATTR(f16_t, 256, 2, 1); NS2VC_RC_ATTR(f16_t, 256, 6, 1);
  NS2VC_RC_ATTR(bf16_t, 128, 1, 2); NS2VC_RC_ATTR(bf16_t, 128, 3, 2); NS2VC_RC_ATTR(f16_t, 128, 1, 2); NS2VC_RC_ATTR(f16_t, 128, 3, 2);
  NS2VC_RC_ATTR(bf16_t, 384, 3, 1); NS2VC_RC_ATTR(bf16_t, 384, 9, 1); NS2VC_RC_ATTR(f16_t, 384, 3, 1); NS2VC_RC_ATTR(f16_t, 384, 9, 1);
  NS2VC_RC_ATTR(bf16_t, 384, 2, 1); NS2VC_RC_ATTR(bf16_t, 384, 5, 1); NS2VC_RC_ATTR(f16_t, 384, 2, 1); NS2VC_RC_ATTR(f16_t, 384, 5, 1);
#undef NS2VC_RC_ATTR
  return hipSuccess;
}

}  // namespace ns2vc
