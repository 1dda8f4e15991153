// Token-stationary GEGLU projection of a transformer block of the NS2VC denoiser, CDNA4 (gfx950), 16-bit operand types.  Round 5.
//
// Replaces, for the dim-384 blocks (level 2: 5 launches of 37 us, the largest plain GEMMs left in the step), the GEMM behind
//     h = (n W1v^T + b1v) * gelu(n W1g^T + b1g),   n = LayerNorm(y)        (reference unet1d/attention.py:178-203, 206-301: ff.net.0, erf GELU)
// with the LayerNorm applied by linearity (gamma / beta folded into W1 / b1 at pack time, as everywhere in this engine).
//
// As a GEMM (M = 7520 rows, N = 3072 packed columns, K = 384) it ran 1416 tiles of 128 x 128 with SIX K tiles each: every tile streams its
// activation rows and its weight rows (192 KB per 12.6 MFLOP), and set-up and epilogue are paid per tile.  K is short enough to turn that around:
// 128 token rows x 384 channels are 96 KB -- 96 VGPRs per lane as MFMA fragments.  One workgroup keeps its tokens IN REGISTERS and sweeps a QUARTER
// of the hidden units, streaming only weight tiles ([128 rows][64 k] = 16 KB, one flat pre-swizzled sequence in consumption order, lane-linear
// LDS-DMA with all addresses in SGPRs, a ring of nine: eight tiles = 128 KB in flight per CU): 590 KB of weights + 96 KB of rows per 75 MFLOP,
// half the bytes of the GEMM, 59 x 4 = 236 workgroups = one round on 256 CUs.
//
// Everything is computed TRANSPOSED, as in ffn.hip (S^T = W1 y^T): after the 32 x 32 MFMA a lane holds 16 hidden units of ONE token (column =
// lane & 31), value and gate of a unit in the same lane and register, so the LayerNorm fix-up is per lane, GEGLU is register arithmetic, and --
// the weight rows of every 32-unit group being stored in the order unit(m) = 16 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3) -- the lane's 16 results are
// 16 CONSECUTIVE hidden units: two 16-byte stores, no LDS staging, no epilogue.  512 threads = 8 waves = 4 token quarters x 2 unit groups of a tile.
#include "common.h"
#include "mma.h"
#include <type_traits>
#include <vector>

#ifndef NS2VC_GG_ABLATE
#define NS2VC_GG_ABLATE 0   // diagnostic builds (wrong results, timing only): 1 no erf / GEGLU product, 2 no MFMAs, 4 no DMA inside the loop, 8 no fragment reads inside the loop
#endif

#ifndef NS2VC_GEMM_TRACE
#define NS2VC_GEMM_TRACE 0
#endif
#ifndef NS2VC_GG_PRIO
#define NS2VC_GG_PRIO 0
#endif
#ifndef NS2VC_GG_ROTATE
#define NS2VC_GG_ROTATE 0     // 1: token block tb sweeps its unit blocks starting at block tb % UB (de-synchronises the L2 requests of an XCD's workgroups; measured: no gain -- the limit is the L2's aggregate rate, not a hot spot)
#endif
#ifndef NS2VC_GG_PINGPONG
#define NS2VC_GG_PINGPONG 1   // 1: the two waves of a SIMD alternate between a load segment and a compute segment, half a step apart; 0: all eight waves in lock step
#endif

namespace ns2vc {

typedef ::ns2vc_geglu_args GegluArgs;

// optional per-wave phase timing (cycle counter): [block][wave][16] = entry, prologue end, then sums over the steps: wait for the tile (vmcnt),
// barrier, tile issue + fragment reads issued, MFMAs + GEGLU chunk (to completion), stores; exit.  Only in trace builds (`make TRACE=1`), set through
// ns2vc_debug_set_gemm_trace, read by tools/geglu_trace.py
__device__ unsigned long long* g_gg_trace = nullptr;
void set_gg_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gg_trace), &p, sizeof(p)); }
#if NS2VC_GEMM_TRACE
#define GG_NOW() __builtin_readcyclecounter()
#define GG_ACC(i) do { const unsigned long long t_ = GG_NOW(); tacc[i] += t_ - tmark; tmark = t_; } while (0)
#else
#define GG_NOW() 0ull
#define GG_ACC(i) do {} while (0)
#endif

constexpr int GG_TILE = 128 * 128;      // bytes: [128 rows][128 B of K]
constexpr int GG_TOK = 128;             // tokens per workgroup
constexpr int GG_SPLIT = 4;             // workgroups per token block (hidden-unit quarters)
constexpr int GG_RING = 8;              // weight tiles resident in LDS (seven in flight: the stream is latency-bound, bytes in flight per CU set its rate)

template <int D> struct GegluGeom {
  static constexpr int KT = D / 64;                          // K tiles
  static constexpr int UB = 4 * D / 64 / GG_SPLIT;           // 64-unit blocks per workgroup (each = one 128-row tile per K tile)
  static constexpr int NP = UB * KT;                         // weight tiles per workgroup
  static constexpr int CONSTS = UB * 128 * 8;                // bytes: (rowsum, bias) per stream row of this workgroup's quarter
  static constexpr int STAGE = 4 * 32 * 128;                 // bytes: result staging, [token quarter][32 tokens][64 units] per unit block
  static constexpr int LDS = GG_RING * GG_TILE + CONSTS + STAGE;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert((4 * D / 64) % GG_SPLIT == 0, "whole unit blocks per quarter");
  static_assert(KT % 2 == 0, "fragment double buffer: the parity of a step is the parity of its K tile");
  static_assert(NP >= GG_RING, "prologue issues a full ring");
};

// unit (0..31) of a 32-unit group that sits at MFMA row m: lane half `hi` then holds units 16 hi .. 16 hi + 15 in register order
static inline int gg_unit_of_row(int m) { return 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3); }

// s_waitcnt vmcnt(2 t) for a run-time t in [0, N] (an immediate in the instruction: a short scalar compare chain, first compare taken in the steady state)
template <int N> __device__ __forceinline__ void gg_wait_tiles(int t) {
  if constexpr (N == 0) wait_vmcnt<0>();
  else { if (t >= N) wait_vmcnt<2 * N>(); else gg_wait_tiles<N - 1>(t); }
}

template <typename TM, int D>
__global__ __launch_bounds__(512) void geglu_kernel(const GegluArgs a) {
  op_mode_init<TM>();
  using G = GegluGeom<D>;
  constexpr int KT = G::KT, UB = G::UB, NP = G::NP, RING = GG_RING;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  const float* const consts = reinterpret_cast<const float*>(smem + RING * GG_TILE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hg = wave & 1, tq = wave >> 1;          // unit group of a tile, token quarter
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmark = 0;
  (void)tacc; (void)tmark;
  const unsigned long long t_entry = GG_NOW();
  (void)t_entry;
  const int l31 = lane & 31, hi = lane >> 5;
  const int sw = (l31 >> 1) & 7;                    // read-side XOR swizzle of every fragment row this lane touches
  const unsigned lds0 = (unsigned)(size_t)smem;

  // ---- XCD-aware order: the four quarters of a token block get consecutive positions on ONE XCD (workgroup ids 8 apart): its token rows are
  // fetched into that L2 once, and every XCD streams the whole weight matrix (2.4 MB) through its L2 for its seven or eight token blocks
  const int ntb = (a.M + GG_TOK - 1) / GG_TOK;
  int tb, q;
  {
    const int nwg = ntb * GG_SPLIT, bid = blockIdx.x;
    const int qn = nwg >> 3, rn = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
    tb = swz / GG_SPLIT;
    q = swz - tb * GG_SPLIT;
  }
  const int m0 = tb * GG_TOK;
  const int mtok = m0 + 32 * tq + l31;              // this lane's token (both lane halves)

  const i32x4_t rW = make_rsrc(a.wstream, (unsigned long long)GG_SPLIT * NP * GG_TILE);
  const unsigned lane16 = (unsigned)(lane * 16);
  const unsigned wq0 = (unsigned)(q * NP) * GG_TILE;        // this quarter's tiles inside the stream
  // a tile = 16 pieces of 1 KB: two per wave, all addresses scalar
#if NS2VC_GG_ROTATE
  // the unit blocks of a quarter are independent: token block tb starts its sweep at unit block tb % UB, so the seven or eight workgroups of an XCD that
  // share a quarter do not all ask its L2 for the same tile at the same moment (results unchanged: only the order of the blocks differs)
  const int rot = __builtin_amdgcn_readfirstlane(tb % UB);
#else
  const int rot = 0;
#endif
  auto ubm = [&](int u) __attribute__((always_inline)) { const int v = u + rot; return v >= UB ? v - UB : v; };     // processing position -> unit block
  auto issue_tile = [&](int p, int slot) __attribute__((always_inline)) {
    const int u = p / KT, pt = ubm(u) * KT + (p - u * KT);
#pragma unroll
    for (int j = 0; j < 2; ++j) blds16(rW, lane16, wq0 + (unsigned)(pt * GG_TILE + (j * 8 + wave) * 1024), lds0 + slot * GG_TILE + (j * 8 + wave) * 1024);
  };
  // ---- DMA: the token rows (96 KB = ring slots 3 .. 8 for now: whole 128-byte rows, each fetched once per workgroup; source-side swizzle, rows
  // past M read as zeros), the constants, the first three weight tiles
  constexpr int NT0 = RING - KT * GG_TOK * 128 / GG_TILE;     // weight tiles that fit beside the token panel at the start
  static_assert(NT0 >= 1, "room for a first weight tile beside the token panel");
  constexpr int PANEL0 = NT0 * GG_TILE;
  static_assert(PANEL0 + KT * GG_TOK * 128 <= RING * GG_TILE, "the token panel fits in the ring slots it borrows");
  {
    const i32x4_t rY = make_rsrc(a.yn, (unsigned long long)a.M * a.ldy * 2ull);
#pragma unroll
    for (int hp2 = 0; hp2 < 2; ++hp2) {               // two passes of 64 rows (8 waves x 8 rows)
      const int prow = 64 * hp2 + 8 * wave + (lane >> 3), pchunk = lane & 7;
      const int m = m0 + prow;
      const unsigned voff = m < a.M ? (unsigned)m * (unsigned)a.ldy * 2u + (unsigned)((pchunk ^ ((prow >> 1) & 7)) * 16) : DMA_OOB;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) blds16(rY, voff, (unsigned)(kt * 128), lds0 + PANEL0 + kt * (GG_TOK * 128) + (64 * hp2 + 8 * wave) * 128);
    }
    const i32x4_t rC = make_rsrc(a.consts, (unsigned long long)GG_SPLIT * G::CONSTS);
#pragma unroll
    for (int j = 0; j < (G::CONSTS + 8191) / 8192; ++j) {
      const unsigned off = (unsigned)((j * 8 + wave) * 1024);
      if (off < (unsigned)G::CONSTS) blds16(rC, lane16, (unsigned)(q * G::CONSTS) + off, lds0 + RING * GG_TILE + off);
    }
  }
#pragma unroll
  for (int p0 = 0; p0 < NT0; ++p0) issue_tile(p0, p0);

  // ---- LayerNorm statistics of this lane's token (ordinary loads: the compiler waits for them -- and, not seeing the DMA above, for everything
  // issued so far: that is the wait for the panel)
  float mean = 0.f, rstd = 1.f;
  {
    const float4* sp = reinterpret_cast<const float4*>(a.ln_stats + (size_t)min(mtok, a.M - 1) * (D / 64) * 2);
    float s = 0.f, qq = 0.f;
#pragma unroll
    for (int i = 0; i < D / 128; ++i) { const float4 v = sp[i]; s += v.x + v.z; qq += v.y + v.w; }
    const float inv = 1.0f / (float)D;
    mean = s * inv;
    double var = (double)qq * (double)inv - (double)mean * (double)mean;
    if (var < 0.0) var = 0.0;
    rstd = 1.0f / sqrtf((float)var + a.ln_eps);
    if (a.ln_health && hg == 0 && q == 0) {          // same health report as the LayerNorm-consumer GEMMs (gemm.hip ln_row_finish)
      float ratio = mtok < a.M ? fabsf(mean) * rstd : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
      if (lane == 0 && ratio > __uint_as_float(__hip_atomic_load(a.ln_health, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        atomicMax(a.ln_health, __float_as_uint(ratio));
    }
  }
  asm volatile("" : "+v"(mean), "+v"(rstd));
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();                     // everyone's pieces of the panel, the constants and tiles 0 .. 2 are in LDS

  // fragments of a weight tile: value rows (0..3) and gate rows (4..7) of this wave's unit group
  auto read_frags = [&](int slot, u32x4_t (&fw)[8]) __attribute__((always_inline)) {
    const char* wrow = ring + slot * GG_TILE + (64 * hg + l31) * 128;       // value row; gate row = + 32 rows
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ((2 * ks + hi) ^ sw) * 16;
      fw[ks] = *reinterpret_cast<const u32x4_t*>(wrow + c);
      fw[4 + ks] = *reinterpret_cast<const u32x4_t*>(wrow + 32 * 128 + c);
    }
  };
  // ---- this lane's token row as MFMA B fragments, in REGISTERS for the rest of the kernel (KT x 4 x 16 B = 96 VGPRs): the K panel is the stationary
  // operand, and from here on LDS holds nothing but streamed weights -- a ring of nine tiles, eight of them in flight
  u32x4_t tk[KT][4];
  u32x4_t fw[2][8];                                 // fragment double buffer: tile p + 1 is read while the MFMAs of tile p run
  {
    const char* trow = smem + PANEL0 + (32 * tq + l31) * 128;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) tk[kt][ks] = *reinterpret_cast<const u32x4_t*>(trow + kt * (GG_TOK * 128) + ((2 * ks + hi) ^ sw) * 16);
#if !NS2VC_GG_PINGPONG
    read_frags(0, fw[0]);
#endif
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(tk[kt][ks]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // everyone has its tokens: the panel's slots join the ring
#pragma unroll
    for (int p0 = NT0; p0 < RING; ++p0) issue_tile(p0, p0);
  }
#if NS2VC_GG_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);     // static priority for the second-dispatched half (the arbitration loser on every step otherwise)
#endif
  const unsigned long long t_pro = GG_NOW();
  (void)t_pro;
  tmark = t_pro;

  int p = 0, slot = 0;                              // tile being multiplied, its ring slot
  f32x16_t av, ag, pv, pg;                          // accumulators of the unit block being multiplied; of the previous one, waiting for its GEGLU
  uint32_t hp[8];                                   // packed results of the previous unit block
#pragma unroll
  for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; pv[r] = 0.f; pg[r] = 0.f; }
#pragma unroll
  for (int r = 0; r < 8; ++r) hp[r] = 0u;

  // One step = one weight tile.  The GEGLU of a unit block (LayerNorm fix-up, bias, erf: ~280 VALU instructions per wave, as much SIMD time as the
  // block's 48 MFMAs take on the matrix pipe) is SOFTWARE-PIPELINED against the MFMAs of the next block: all eight waves move in lock step (a
  // barrier per tile), so nothing else would ever overlap the two -- a quarter of it rides in each of the first four steps, the two stores in the fifth.
  //
  // Waits: at step p the fragments of tile p are in registers and tile p + 1 must have landed for everyone.  Of what this wave issued after it, the
  // RING - 2 tiles p + 2 .. p + RING - 1 (two pieces each) may still be in flight.  vmcnt also counts the two result stores per unit block; "at most 2 t operations outstanding" stays the condition: loads
  // complete in order among themselves, so a piece of tile p + 1 can only be outstanding together with all 2 t younger pieces -- more than 2 t
  // operations -- whatever the stores do (they can only make the wait longer).
  auto chunk = [&](int ubp, int j) __attribute__((always_inline)) {
    // register r <-> stream row (r&3) + 8 (r>>2) + 4 hi of this wave's 32 = unit 16 hi + r of the group; this is r = 4 j .. 4 j + 3
    const float* cv = consts + (size_t)(128 * ubm(ubp) + 64 * hg) * 2;      // (rowsum, bias) of the value rows; gate rows = + 32
    const float4 v0 = *reinterpret_cast<const float4*>(cv + (8 * j + 4 * hi) * 2);
    const float4 v1 = *reinterpret_cast<const float4*>(cv + (8 * j + 4 * hi) * 2 + 4);
    const float4 g0 = *reinterpret_cast<const float4*>(cv + (32 + 8 * j + 4 * hi) * 2);
    const float4 g1 = *reinterpret_cast<const float4*>(cv + (32 + 8 * j + 4 * hi) * 2 + 4);
    const float wsv[4] = {v0.x, v0.z, v1.x, v1.z}, bv[4] = {v0.y, v0.w, v1.y, v1.w};
    const float wsg[4] = {g0.x, g0.z, g1.x, g1.z}, bg[4] = {g0.y, g0.w, g1.y, g1.w};
    float hh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xv = rstd * (pv[4 * j + i] - mean * wsv[i]) + bv[i];
      const float xg = rstd * (pg[4 * j + i] - mean * wsg[i]) + bg[i];
#if NS2VC_GG_ABLATE & 1
      hh[i] = xv + xg;
#else
      hh[i] = xv * gelu_erf_f(xg);
#endif
    }
    hp[2 * j] = Op16<TM>::pack(hh[0], hh[1]);
    hp[2 * j + 1] = Op16<TM>::pack(hh[2], hh[3]);
  };
  // Result stores.  A lane holds 16 units of ONE token: stored from the registers, every instruction is 64 scattered 16-byte pieces in 32 rows, and the
  // store path -- not the MFMAs, not the weight stream -- set the kernel's time (with everything else removed from the loop it still ran 25 of its
  // 38 us).  So a unit block goes through LDS: the two waves of a token quarter write their halves of [32 tokens][64 units] (swizzled 16-byte chunks),
  // and after the next barrier each stores 16 of the rows: two instructions of 8 full 128-byte lines.  Buffer stores: descriptor in SGPRs, one
  // 32-bit offset register per lane; a row past M carries an out-of-range offset (the hardware drops it), so every wave issues the same stores.
  char* const stage = smem + RING * GG_TILE + G::CONSTS + tq * (32 * 128);
  const i32x4_t rO = uniform_rsrc(make_rsrc(a.out_op, (unsigned long long)a.M * (unsigned long long)a.ldo * 2ull));
  const int srow = 16 * hg + (lane >> 3);           // staged row this lane stores (and + 8)
  unsigned ooff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + 32 * tq + srow + 8 * j;
    ooff[j] = m < a.M ? ((unsigned)m * (unsigned)a.ldo + (unsigned)(q * UB * 64)) * 2u + (unsigned)((lane & 7) * 16) : 0xC0000000u;
  }
  auto stage_block = [&]() __attribute__((always_inline)) {
    // units 32 hg + 16 hi .. + 15 of token l31: chunks 4 hg + 2 hi, + 1 of its row
    char* row = stage + l31 * 128;
    const int c0 = 4 * hg + 2 * hi, x = l31 & 7;
    const u32x4_t v0 = {hp[0], hp[1], hp[2], hp[3]}, v1 = {hp[4], hp[5], hp[6], hp[7]};
    *reinterpret_cast<u32x4_t*>(row + ((c0 ^ x) * 16)) = v0;
    *reinterpret_cast<u32x4_t*>(row + (((c0 + 1) ^ x) * 16)) = v1;
  };
  auto store_block = [&](int ubp) __attribute__((always_inline)) {
    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(ubm(ubp) * 128);
    const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(stage + srow * 128 + (((lane & 7) ^ (srow & 7)) * 16));
    const u32x4_t v1 = *reinterpret_cast<const u32x4_t*>(stage + (srow + 8) * 128 + (((lane & 7) ^ (srow & 7)) * 16));
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen " NS2VC_WT_MOD "\n\tbuffer_store_dwordx4 %4, %5, %2, %3 offen " NS2VC_WT_MOD "\n\ts_nop 1"
                 :: "v"(v0), "v"(ooff[0]), "s"(rO), "s"(so), "v"(v1), "v"(ooff[1]) : "memory");
  };
  // MM: a tile is multiplied in this step; EP: a quarter of the previous unit block's GEGLU rides along
#if NS2VC_GG_PINGPONG
  // PING-PONG.  Lock-stepped, the two waves of a SIMD want the matrix pipe at the same time and the DMA / LDS / store paths at the same time: measured,
  // the parts of a step ADD UP (step = MFMAs + fragment reads + DMA issue + GEGLU, ~2000 cycles) instead of overlapping.  So waves 4 .. 7 (the SIMD
  // partners of waves 0 .. 3) run HALF A STEP BEHIND: every step is a load segment (refill a ring slot, read this tile's fragments, stores) and a
  // compute segment (8 MFMAs with a quarter of the previous block's GEGLU between them), a barrier after each, and one extra barrier at the start
  // of the late half / the end of the early half.  While one wave of a SIMD multiplies, its partner loads.
  //   early half:  L0 | C0 | L1 | C1 | ...            late half:  -- | L0 | C0 | L1 | ...
  // Tile p is read by the early half in tick 2p and by the late half in tick 2p + 1: its slot is refilled (with tile p + RING) from tick 2p + 2 on,
  // i.e. in the load segment of step p + 1; and it has landed for everyone before tick 2p: every wave waits for its own pieces of tile p + 1 at the
  // END of its load segment p (for the late half that is the last barrier before the early half reads them).
  auto step = [&](auto mm_, auto ep_, int kt_, int ub) __attribute__((always_inline)) {
    constexpr bool MM = decltype(mm_)::value, EP = decltype(ep_)::value;
    const int kt = kt_;
    if constexpr (MM) {
#if !(NS2VC_GG_ABLATE & 4)
      if (p >= 1 && p - 1 + RING < NP) issue_tile(p - 1 + RING, slot == 0 ? RING - 1 : slot - 1);
#endif
#if NS2VC_GG_ABLATE & 8
      if (p == 0)
#endif
      read_frags(slot, fw[0]);
      GG_ACC(2);
      if constexpr (EP) { if (kt == 4) { store_block(ub - 1); GG_ACC(4); } }
#if !(NS2VC_GG_ABLATE & 4)
      if (p + 1 < NP) gg_wait_tiles<RING - 2>(min(RING - 2, NP - 2 - p));
#endif
      GG_ACC(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragments are in registers
      __builtin_amdgcn_s_barrier();
      GG_ACC(1);
      if constexpr (EP) { if (kt < 4) chunk(ub - 1, kt); }
#if !(NS2VC_GG_ABLATE & 2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        MmaT<TM>::mma(av, fw[0][ks], tk[kt][ks]);
        MmaT<TM>::mma(ag, fw[0][4 + ks], tk[kt][ks]);
      }
#endif
      if constexpr (EP) {
        if (kt < 4) asm volatile("" : "+v"(hp[2 * kt]), "+v"(hp[2 * kt + 1]));     // computed in THIS segment (not sunk to the store)
        if (kt == 3) { stage_block(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }    // staged before the barrier, stored after it (load segment of step 4)
      }
      if (NS2VC_GEMM_TRACE) { asm volatile("" : "+v"(av), "+v"(ag)); }
      GG_ACC(3);
      __builtin_amdgcn_s_barrier();
      GG_ACC(1);
      ++p;
      slot = slot + 1 == RING ? 0 : slot + 1;
    } else {
      if constexpr (EP) {
        if (kt < 4) chunk(ub - 1, kt);
        if (kt == 3) { stage_block(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // (the tail: one barrier, the same for everyone)
        if (kt == 4) store_block(ub - 1);
      }
    }
  };
  if (wave >= 4) __builtin_amdgcn_s_barrier();      // the late half starts one tick later ...
#else
  auto step = [&](auto mm_, auto ep_, int kt_, int ub) __attribute__((always_inline)) {
    constexpr bool MM = decltype(mm_)::value, EP = decltype(ep_)::value;
    const int kt = kt_;
    if constexpr (MM) {
      const int nslot = slot + 1 == RING ? 0 : slot + 1;
#if !(NS2VC_GG_ABLATE & 4)
      if (p + 1 < NP) gg_wait_tiles<RING - 2>(min(RING - 2, NP - 2 - p));
#endif
      GG_ACC(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads of tile p (issued in step p - 1) are done: its slot may be refilled
      __builtin_amdgcn_s_barrier();
      GG_ACC(1);
#if !(NS2VC_GG_ABLATE & 4)
      if (p + RING < NP) issue_tile(p + RING, slot);
#endif
#if !(NS2VC_GG_ABLATE & 8)
      if (p + 1 < NP) read_frags(nslot, fw[(kt + 1) & 1]);
#endif
      GG_ACC(2);
      slot = nslot;
    }
    if constexpr (EP) { if (kt < 4) chunk(ub - 1, kt); }
    if constexpr (MM) {
#if !(NS2VC_GG_ABLATE & 2)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        MmaT<TM>::mma(av, fw[kt & 1][ks], tk[kt][ks]);
        MmaT<TM>::mma(ag, fw[kt & 1][4 + ks], tk[kt][ks]);
      }
#endif
      ++p;
    }
    if constexpr (EP) {
      if (kt < 4) asm volatile("" : "+v"(hp[2 * kt]), "+v"(hp[2 * kt + 1]));     // computed in THIS step (not sunk to the store four steps later)
      if (kt == 3) {
        stage_block();                                // (step 4 begins with lgkmcnt(0) and a barrier)
        if constexpr (!MM) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
      }
    }
    if (NS2VC_GEMM_TRACE) { asm volatile("" : "+v"(av), "+v"(ag)); GG_ACC(3); }
    if constexpr (EP) {
      if (kt == 4) { store_block(ub - 1); GG_ACC(4); }
    }
  };
#endif
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) step(T_{}, F_{}, kt, 0);
#pragma unroll 1
  for (int ub = 1; ub < UB; ++ub) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { pv[r] = av[r]; pg[r] = ag[r]; av[r] = 0.f; ag[r] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) step(T_{}, T_{}, kt, ub);
  }
#if NS2VC_GG_PINGPONG
  if (wave < 4) __builtin_amdgcn_s_barrier();       // ... and the early half waits for it once at the end: the same number of barriers for everyone
#endif
#pragma unroll
  for (int r = 0; r < 16; ++r) { pv[r] = av[r]; pg[r] = ag[r]; }
#pragma unroll
  for (int kt = 0; kt < 5; ++kt) step(F_{}, T_{}, kt, UB);
#if NS2VC_GEMM_TRACE
  if (g_gg_trace && lane == 0) {
    unsigned long long* tr = g_gg_trace + ((size_t)blockIdx.x * 8 + wave) * 16;
    tr[0] = t_entry; tr[1] = t_pro;
    for (int i = 0; i < 5; ++i) tr[2 + i] = tacc[i];
    tr[7] = GG_NOW();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    tr[8] = hwid;                                   // bits 0-3 wave slot, 4-5 SIMD, 8-11 CU, ...
  }
#endif
}

// ---------------------------------------------------------------------------
// host side: tile-stream packer and launcher
// ---------------------------------------------------------------------------
bool geglu_eligible(int dim, int T, int prec) { return dim == 384 && T >= 1 && (prec == PREC_BF16 || prec == PREC_F16); }

// w1p [8 dim][dim]: LayerNorm-folded ff.net.0 rows in the packed (32 value | 32 gate) order of the GEGLU GEMM (engine.cpp pack_all / ns2vc_pack_ffn); bias1p [8 dim].
// Output: the tile stream ([quarter][unit block][K tile][128 rows][64 k], pre-swizzled) and the constants ([quarter][stream row] (rowsum of the ROUNDED row, bias))
// in stream row order, i.e. with the rows of every 32-unit group permuted by gg_unit_of_row.
hipError_t pack_geglu_stream(const float* w1p, const float* bias1p, int dim, int prec, std::vector<unsigned short>& stream, std::vector<float>& consts) {
  if (dim != 384 || (prec != PREC_BF16 && prec != PREC_F16)) return hipErrorInvalidValue;
  const int KT = dim / 64, NUB = 4 * dim / 64;                 // all unit blocks (128 stream rows each)
  stream.clear();
  stream.reserve((size_t)8 * dim * dim);
  consts.assign((size_t)8 * dim * 2, 0.f);
  auto src_row = [&](int sr) {                                  // packed row behind stream row sr
    const int grp = sr >> 6, half = (sr >> 5) & 1, m = sr & 31;
    return grp * 64 + half * 32 + gg_unit_of_row(m);
  };
  for (int ubg = 0; ubg < NUB; ++ubg)
    for (int kt = 0; kt < KT; ++kt)
      for (int r = 0; r < 128; ++r) {
        const int pr = src_row(ubg * 128 + r);
        for (int pos = 0; pos < 8; ++pos) {
          const int lc = pos ^ ((r >> 1) & 7);
          for (int e = 0; e < 8; ++e) stream.push_back(f32_to_op16_bits(w1p[(size_t)pr * dim + 64 * kt + lc * 8 + e], prec));
        }
      }
  for (int sr = 0; sr < 8 * dim; ++sr) {
    const int pr = src_row(sr);
    double s = 0.0;
    for (int k = 0; k < dim; ++k) s += (double)op16_bits_to_f32(f32_to_op16_bits(w1p[(size_t)pr * dim + k], prec), prec);
    consts[2 * sr] = (float)s;
    consts[2 * sr + 1] = bias1p ? bias1p[pr] : 0.f;
  }
  return hipSuccess;
}

template <typename TM> static hipError_t launch_geglu_t(const GegluArgs& a, hipStream_t s) {
  const int ntb = (a.M + GG_TOK - 1) / GG_TOK;
  hipLaunchKernelGGL((geglu_kernel<TM, 384>), dim3(ntb * GG_SPLIT), dim3(512), GegluGeom<384>::LDS, s, a);
  return hipGetLastError();
}

hipError_t launch_geglu(const GegluArgs& a, int prec, hipStream_t s) {
  if (!geglu_eligible(a.dim, 1, prec) || a.M <= 0) return hipErrorInvalidValue;
  if (!a.yn || !a.ln_stats || !a.wstream || !a.consts || !a.out_op) return hipErrorInvalidValue;
  if (a.ldy < a.dim || a.ldo < 4 * a.dim || (a.ldy & 7) || (a.ldo & 7) || (unsigned long long)a.M * a.ldy * 2ull > 0xFFF00000ull || (unsigned long long)a.M * a.ldo * 2ull > 0x7FF00000ull) return hipErrorInvalidValue;
  if ((reinterpret_cast<uintptr_t>(a.out_op) & 15) != 0) return hipErrorInvalidValue;
  return prec == PREC_BF16 ? launch_geglu_t<bf16_t>(a, s) : launch_geglu_t<f16_t>(a, s);
}

hipError_t init_geglu_attributes() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(geglu_kernel<bf16_t, 384>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GegluGeom<384>::LDS);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(geglu_kernel<f16_t, 384>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)GegluGeom<384>::LDS);
  return e;
}

}  // namespace ns2vc
