// Tap-sharing implicit-GEMM kernel for the k = 3, stride-1 Conv1d of the NS2VC denoiser, CDNA4 (gfx950).  Round 5.
//
// Replaces gemm4_kernel for the 46 `F.conv1d(k=3, padding=1)` call sites of a step (unet1d/resnet.py:591-641 conv1 / conv2 of
// every ResnetBlock2D incl. the concat inputs of the up blocks and the fused 1x1 shortcut; unet_1d_condition.py:943,1032
// conv_in / conv_out) -- 35 % of the step's time in round 4, at 12 % MFMA utilisation.
//
// What bounded gemm4_kernel there (profiles/r03_gemm_spec.txt, r04_pmc_*): its K loop streams (BM + BN) x 128 B per 64-wide
// K tile from L2 into LDS -- 24.5 KB per 1.05 MFLOP with the 64 x 128 tiles the coarse levels need to fill the chip -- at the
// ~31 B/clk/CU the L2 -> LDS path delivers, against the 96 B/clk the MFMAs of such a tile could eat.  For a k = 3 convolution a
// third of those bytes is the SAME activation rows fetched three times: tap tau of output row r reads input row r + tau - 1, so
// the three taps' A tiles are one another shifted by a row.  This kernel loads an activation chunk ONCE:
//
//   * rows live in a PADDED row space of period P = T + 1 per batch item (index q = b P + t; t == T is a pad row: its input is
//     zero -- an out-of-range DMA offset -- and its output is never stored), so "the row above / below" is always q -/+ 1 and
//     the zero padding at both ends of every item is simply there;
//   * a tile is 128 panel rows x BN columns: the panel holds padded input rows q0-1 .. q0+126 of one 64-channel chunk (16 KB),
//     the MFMAs of tap tau read panel rows r + tau, r = 0 .. 127, and the tile OWNS output rows q0 .. q0+125 (rows 126 / 127
//     read two rows past the panel: computed, never stored) -- 126 / 128 of the MFMA work is useful, no halo DMA pass;
//   * per 64-channel chunk: ONE 16-KB activation chunk + three (BN x 128 B) weight tiles (k offset tau * Ctot + c: the packed
//     [N][K] layout already has them, just in another order).  BN = 64: 13.3 KB per 1.05 MFLOP (gemm4: 24.5), BN = 128:
//     21.3 KB per 2.1 MFLOP (gemm4 128 x 128: 32) -- and the 128 x 64 tile gives as many workgroups as gemm4's 64 x 128;
//   * activation chunks and weight tiles ride separate 3-deep rings (a chunk is issued two chunks ahead, a weight tile two
//     steps ahead), counted s_waitcnt vmcnt, one s_barrier per step; loader / consumer wave specialisation as in gemm4's
//     SPEC kernels: NL loader waves issue every LDS-DMA piece, 4 consumer waves own the MFMAs;
//   * the concat of the up blocks is a chunk walk over two descriptors, the fused 1x1 shortcut (K segment c2) a run of
//     single-tap chunks (tau = 1: the centre rows) behind the main ones;
//   * the GroupNorm-apply prologue (gnpro.h) builds exactly the real rows the panel will read; its cooperative form shares them
//     between the N / BN column tiles of a row block;
//   * epilogue as gemm4's (LDS-staged transpose, whole-row stores, bias, fp32 residual, fp32 + operand stores, int64 fixed-point
//     GroupNorm statistics), with the padded -> real row map applied per stored row.
//
// The summation order over K differs from gemm4_kernel's (chunk-major instead of tap-major), so results agree with it to fp32
// rounding, not bitwise; within this kernel everything is deterministic.
#include "common.h"
#include "mma.h"
#include "gnpro.h"
#include <type_traits>
#include <vector>
#include <cstring>

namespace ns2vc {

constexpr int TS_ROW = 128;    // bytes of K per tile row (64 x 16 bit / 32 fp32)
constexpr int TS_BM = 128;     // panel rows
constexpr int TS_BMO = 126;    // output rows a tile owns
constexpr int TS_ASLOT = TS_BM * TS_ROW;

// optional per-workgroup phase timing (s_memtime; compiled in only with -DNS2VC_GEMM_TRACE=1, `make TRACE=1`; tools/ts_trace.py):
// [block][16] u64: 0 entry, 7 prologue done, 1 loop begin, 2 loop end, 3 exit | loader wave 0: 4 sum of counted waits, 5 sum of barrier waits, 6 sum of issue
// time | first consumer wave: 8 sum of barrier waits (incl. its own LDS drain), 9 sum of read + MFMA time
__device__ unsigned long long* g_ts_trace = nullptr;
void set_ts_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ts_trace), &p, sizeof(p)); }
#ifndef NS2VC_GEMM_TRACE
#define NS2VC_GEMM_TRACE 0
#endif
#if NS2VC_GEMM_TRACE
#define TS_NOW() __builtin_readcyclecounter()
#define TS_STAMP(i) do { if (tr && (tid == 0)) tr[i] = TS_NOW(); } while (0)
#define TS_CLK(t) do { t = TS_NOW(); } while (0)
#define TS_ACC(a, t) do { const unsigned long long n_ = TS_NOW(); a += n_ - t; t = n_; } while (0)
#define TS_PUT(i, v) do { if (tr && (lane == 0) && (wave == 0 || wave == NL)) tr[i] = (v); } while (0)
#else
#define TS_STAMP(i) do { (void)tr; } while (0)
#define TS_CLK(t) do {} while (0)
#define TS_ACC(a, t) do {} while (0)
#define TS_PUT(i, v) do {} while (0)
#endif
#ifndef NS2VC_TS_ABLATE
#define NS2VC_TS_ABLATE 0        // diagnostic builds (wrong results): 1 = consumers neither read fragments nor multiply, 2 = loaders issue no DMA inside the loop,
#endif                           // 4 = consumers read fragments but do not multiply
#ifndef NS2VC_TS_KS_DEFAULT
#define NS2VC_TS_KS_DEFAULT 0    // K-split consumer layout of the 64-column tile by default (set after the same-box A/B)
#endif
#ifndef NS2VC_TS_NL_DEFAULT
#define NS2VC_TS_NL_DEFAULT 8    // loader waves (r5 session 6, tap-granular loop, same box: 3.564 ms/step with 8, 3.576 with 4)
#endif
#ifndef NS2VC_TS_GNP_XB
#define NS2VC_TS_GNP_XB 4        // fp32 rows in flight per thread in the GroupNorm prologue of this kernel (gemm4 keeps gnpro.h's 6).  r6, same box, four runs each, after the
#endif                           // spilled registers were gone: 6 / 5: 3.557-3.568 ms/step, 4: 3.542-3.549 (-0.45 %), 3: 3.540-3.560 (profiles/r06_ab_split_io.txt)
#ifndef NS2VC_TS_NLD4
#define NS2VC_TS_NLD4 0          // four DMA waves in the chunk-granular loop of the eight-loader kernels: faster isolated (L2-warm operands), 0.8 % SLOWER in the step
#endif                           // (3.645 vs 3.617 ms/step same box, profiles/r06_ab_chunk_loop.txt) -- off
#ifndef NS2VC_CONS_PF
#define NS2VC_CONS_PF 1          // consumer waves: every fragment read of a step before its first MFMA (0: the compiler's order)
#endif
// Weight ring depth: a tile is issued SW - 1 steps ahead.  r5 sessions 4 / 5 / 12 (tools/ts_trace.py, profiles/r05_ts_trace.txt, r05_ts_ablate.txt): five steps
// ahead (SW = 6) changed nothing against two (3.62 ms/step either way) -- the loop does not wait for latency: the DMA stream ALONE (consumers idle)
// takes 95 % of the full kernel's time, i.e. the loop runs at the rate the L2 -> LDS path delivers tiles to all CUs at once (~22 B/clk/CU here).  So
// the ring stays as shallow as the pipeline needs (72 / 96 KB: two 128 x 64 workgroups per CU); the counted-wait bookkeeping is written for any depth.
//
// r6 (tools/convoy_probe.hip, profiles/r06_convoy_probe.txt): the DMA-only replay of this loop's address stream runs 1637 / 1843 cycles per 64-channel chunk
// (operands L2-warm / from the Infinity Cache) with one counted wait + one s_barrier per TAP step -- exactly the 262 ns per step of the r5 ablation build -- but
// 783 / 1035 cycles per chunk with ONE wait + barrier per CHUNK (activation chunk and the chunk's three weight tiles issued back to back, two chunks ahead):
// what bounded the K loop was neither the L2 hit rate (a run-ahead L2 prefetch of the XCD's unique lines made the replay 9 % SLOWER) nor an L2 -> LDS
// ceiling, but the number of synchronisation points per byte in flight.  NS2VC_TS_CHUNK = 1 (default) runs the loop chunk-granular: the weight ring holds
// whole chunks (9 tiles = 3 chunks at BN = 64, 6 tiles = 2 chunks at BN = 128: 120 / 144 KB of LDS), the consumers meet the loaders once per chunk and run the
// chunk's taps back to back.  NS2VC_TS_CHUNK = 0 keeps the r5 tap-granular loop for A/B builds.
#ifndef NS2VC_TS_CHUNK
#define NS2VC_TS_CHUNK 1
#endif
#if NS2VC_TS_CHUNK
template <int BN> struct TsRing { static constexpr int SW = BN == 64 ? 9 : 6; };
#else
template <int BN> struct TsRing { static constexpr int SW = 3; };
#endif
// s_waitcnt vmcnt(n) for a wave-uniform n in [LO, HI]: the count is an immediate, so a binary tree of scalar branches picks it
template <int LO, int HI> struct TsWait {
  static __device__ __forceinline__ void run(int n) {
    if constexpr (LO == HI) wait_vmcnt<LO>();
    else { constexpr int MID = (LO + HI + 1) / 2; if (n >= MID) TsWait<MID, HI>::run(n); else TsWait<LO, MID - 1>::run(n); }
  }
};

// KS (BN = 64 only): the four consumer waves as 2 row halves x 2 K halves of 64 x 64 wave tiles instead of four 32 x 64 tiles over the whole K:
// 8 KB instead of 12 KB of fragment reads per wave and step (1 KB per MFMA instead of 1.5); the two K halves meet in the LDS-staged epilogue as
// gemm4_kernel's do.  Measured (r5 session 11): 11.9 vs 11.5 us isolated, 3.543 vs 3.542 ms/step in situ -- the consumers hide under the DMA stream either
// way (profiles/r05_ts_ablate.txt), so it stays a tested option (NS2VC_TS_KS_DEFAULT 0).
// GNP: 0 = no GroupNorm in front, 1 (3: writing hi + lo operand pairs, gnp_pair) = the materialising prologue (gnpro.h GnPrologue: rows written to a0, read back by DMA), 2 = the loader waves
// normalise inside the K loop (gnpro.h GnInloop; chunk-granular loop only)
// SOL: with the solver-update epilogue (GemmArgs.sol_*; its own instantiations: compiled into every kernel it cost the 128-column tiles 26 spilled registers and the
// step 0.2 %, profiles/r06_ab_split_io.txt)
template <typename TM, int BN, int NL, int GNP, bool KS = false, bool SOL = false>
__global__ __launch_bounds__(64 * (NL + 4)) void conv3ts_kernel(const GemmArgs g) {
  op_mode_init<TM>();
  constexpr int EPC = MmaT<TM>::EPC;
  constexpr int BKE = 8 * EPC;                                   // channels per chunk
  constexpr int NW = NL + 4, EOFF = NW - 8;                      // waves; first wave with a role in the 8-wave epilogue
  // GNP == 2: the NL non-consumer waves split into NLD DMA waves (every LDS-DMA piece) and NL - NLD producer waves (the in-loop GroupNorm: only
  // compiler-visible loads, so the compiler's own counted waits are exact there, while the DMA waves' inline-asm loads are counted by hand)
  // NS2VC_TS_NLD4 (diagnostic, off): four DMA waves also when all eight are loaders otherwise.  Isolated, with L2-warm operands, the level-3 convs run
  // 16.5 / 11.2 us with four and 18.4 / 12.1 us with eight DMA waves; inside the step (operands from the Infinity Cache) eight are better.
  constexpr int NLD = (GNP == 2 || (NS2VC_TS_CHUNK && NS2VC_TS_NLD4 && NL == 8)) ? 4 : NL;
  constexpr int LTH = NLD * 64, RPP = LTH / 8, PASSB = RPP * TS_ROW;
  constexpr int LA = TS_BM / RPP, LB = BN / RPP;                 // 16-B DMA pieces per loading thread: activation chunk / weight tile
  static_assert(!KS || BN == 64, "the K-split consumer layout is the 64-column tile's");
  constexpr int WGN = BN / 64, WGM = KS ? 2 : 4 / WGN, WM = TS_BM / WGM, MT = WM / 32, NT = 2;
  constexpr int NKK = KS ? 2 : 4;                                // 32-B k-slabs of a step a consumer wave multiplies
  constexpr int WSLOT = BN * TS_ROW;
  constexpr int SW = TsRing<BN>::SW, D = SW - 1;                 // weight ring: a tile is issued D steps ahead
  constexpr unsigned SZB = sizeof(TM);
  static_assert(BN == 64 || BN == 128, "BN");
  static_assert(LA >= 1 && LB >= 1 && (NS2VC_TS_CHUNK ? SW : D) * LB + 3 * LA <= 60, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const aring = smem;                                      // 3 activation chunks
  char* const wring = smem + 3 * TS_ASLOT;                       // SW weight tiles
  char* const tabmem = smem + 3 * TS_ASLOT + SW * WSLOT;         // GNP == 2: the (mean, rstd) table, kept for the whole loop
  static_assert(GNP != 2 || NS2VC_TS_CHUNK, "the in-loop GroupNorm rides the chunk-granular loop");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave < NLD, consumer = wave >= NL;
  const bool producer = GNP == 2 && wave >= NLD && wave < NL;
  const int ewave = wave - EOFF;                                 // role in the epilogue (< 0: none)
  const int cw = consumer ? wave - NL : 0;                       // consumer wave: its wave tile
  const int wm = KS ? cw >> 1 : cw / WGN, wn = KS ? 0 : cw % WGN;
  const int kh = KS ? cw & 1 : 0;                                // K half of a consumer wave (KS)
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned long long* const tr = (NS2VC_GEMM_TRACE && g_ts_trace) ? g_ts_trace + (size_t)blockIdx.x * 16 : nullptr;
  TS_STAMP(0);

  // ---- tile in the padded row space
  const int T = g.Tin, P = T + 1, MP = g.B * P;
  const int nb_n = g.N / BN;
  const int nb_m = (MP + TS_BMO - 1) / TS_BMO;
  int tm, tn;
  const bool coop = GNP && g.gnp_x != nullptr && g.gnp_sync != nullptr && nb_n > 1;     // (uniform over the grid; the launcher pads the grid for it)
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    if (coop) {           // whole row blocks per XCD: the column tiles that share a row block's rows share an L2
      const int q = nb_m >> 3, r = nb_m & 7;
      const int tml = idx / nb_n;
      if (tml >= q + (xcd < r ? 1 : 0)) return;
      tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + tml;
      tn = idx - tml * nb_n;
    } else {
      const int nwg = nb_n * nb_m;
      const int q = nwg >> 3, r = nwg & 7;
      const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
      tm = swz / nb_n;
      tn = swz - tm * nb_n;
    }
  }
  const int q0 = tm * TS_BMO, n0 = tn * BN;

  // ---- step list: ncm main chunks of three steps (tau = 0, 1, 2), then ncs single-step chunks of the fused 1x1 segment (tau = 1)
  const int Ctot = g.c0 + g.c1;
  const int ncm = Ctot / BKE, ncs = g.c2 / BKE, NCH = ncm + ncs;
  const int K1 = 3 * Ctot;
  const int S = 3 * ncm + ncs;

  // ---- DMA coordinates.  Piece j of a loading thread = panel / weight row j*RPP + tid/8, physical 16-B chunk tid%8 (source-side
  // swizzle: physical chunk c of row r holds logical chunk c ^ ((r>>1)&7)).
  const int prow = (tid & (LTH - 1)) >> 3, pchunk = tid & 7;
  const unsigned acolb = (unsigned)((pchunk ^ ((prow >> 1) & 7)) * EPC) * SZB;
  unsigned vw[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) vw[j] = ((unsigned)(n0 + j * RPP + prow) * (unsigned)g.K) * SZB + acolb;
  const i32x4_t rW = make_rsrc(g.w, (unsigned long long)g.N * g.K * SZB);
  int wc_ch = 0, wc_tau = 0;                                     // issue cursor of the weight stream (step order)
  // Tile-major weights (GemmArgs.w_tiled, pack_conv3_tiled below): the [64 rows][128 B] block of (64-column group, step) is ONE contiguous,
  // pre-swizzled 8 KB, in the order the loop consumes the steps -- a piece is 1 KB of consecutive bytes instead of eight 128-B row segments
  // K * 2 bytes apart.  The tiles come from beyond L2 every step (this layer's weights were last read a step ago; every XCD reads all of
  // them), and DRAM / MALL serve consecutive bytes much better than scattered lines (profiles/r05_ab_weight_tiles.txt).
  const bool wtiled = g.w_tiled != nullptr;
  const i32x4_t rWt = make_rsrc(wtiled ? g.w_tiled : g.w, (unsigned long long)g.N * g.K * SZB);
  int wc_step = 0;
  unsigned vwt[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int r = j * RPP + prow;                                // row inside the BN-row tile
    vwt[j] = (unsigned)(((n0 >> 6) + (r >> 6)) * S) * 8192u + (unsigned)((r & 63) * TS_ROW + pchunk * 16);
  }
  auto issue_w = [&](int slot) __attribute__((always_inline)) {
    const int koff = wc_ch < ncm ? wc_tau * Ctot + wc_ch * BKE : K1 + (wc_ch - ncm) * BKE;
    if (loader) {
      const unsigned base = lds0 + 3 * TS_ASLOT + slot * WSLOT + wave * 1024;
      if (wtiled) {
#pragma unroll
        for (int j = 0; j < LB; ++j) blds16(rWt, vwt[j], (unsigned)wc_step * 8192u, base + j * PASSB);
      } else {
#pragma unroll
        for (int j = 0; j < LB; ++j) blds16(rW, vw[j], (unsigned)koff * SZB, base + j * PASSB);
      }
    }
    ++wc_step;
    if (wc_ch < ncm && wc_tau < 2) ++wc_tau;
    else { ++wc_ch; wc_tau = wc_ch < ncm ? 0 : 1; }
  };
  // the first D weight tiles depend on nothing but the tile's column: in flight while the row offsets (a division per piece) and
  // the GroupNorm prologue are still being worked out
#if NS2VC_TS_CHUNK
  // Chunk-granular schedule, fixed: the weight tiles of chunk c go out DG chunks ahead (DG = 2 with the 9-tile ring, 1 with the 6-tile ring), its rows two
  // chunks ahead; within a step weights first, rows second.  The loader's own instruction stream is what bounded the r5 loop (r6 trace: 140-250 cycles of
  // mostly scalar cursor arithmetic and branches per DMA instruction, so that eight loader waves delivered no more than four): the cursors here are one
  // running tile index (tile-major weights: soffset = index * 8 KB), one ring slot each, and the waits take one of four immediates.
  constexpr int DG = SW / 3 - 1;
  int wtile = 0, wslot = 0;                                      // next weight tile of the stream / its ring slot
  auto issue_wchunk = [&](int c) __attribute__((always_inline)) {  // the tiles of chunk c: three taps, or the single tap of a 1x1-segment chunk
    const int nt = c < ncm ? 3 : 1;
    if (wtiled) {
      if (loader) {
        for (int t = 0; t < nt; ++t) {
          const unsigned base = lds0 + 3 * TS_ASLOT + wslot * WSLOT + wave * 1024;
#pragma unroll
          for (int j = 0; j < LB; ++j) blds16(rWt, vwt[j], (unsigned)wtile * 8192u, base + j * PASSB);
          ++wtile;
          wslot = wslot == SW - 1 ? 0 : wslot + 1;
        }
      }
    } else {
      for (int t = 0; t < nt; ++t) { issue_w(wslot); wslot = wslot == SW - 1 ? 0 : wslot + 1; }
    }
  };
#pragma unroll
  for (int d = 0; d < DG; ++d)
    if (d < NCH) issue_wchunk(d);
#else
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < S) issue_w(d);
#endif

  unsigned off0[LA], off1[LA], off2[LA];
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int qi = q0 - 1 + j * RPP + prow;                      // padded input row of this panel row
    const int b = qi > 0 ? qi / P : 0;
    const int t = qi - b * P;
    const bool ok = qi >= 0 && qi < MP && t < T;
    const unsigned row = (unsigned)(qi - b);                     // = b T + t
    off0[j] = ok ? row * (unsigned)g.lda0 * SZB + acolb : DMA_OOB;
    off1[j] = (ok && g.c1) ? row * (unsigned)g.lda1 * SZB + acolb : DMA_OOB;
    off2[j] = (ok && g.c2) ? row * (unsigned)g.lda2 * SZB + acolb : DMA_OOB;
  }
  const unsigned long long rowsA = (unsigned long long)g.B * T;
  const i32x4_t rA0 = make_rsrc(g.a0, rowsA * g.lda0 * SZB);
  const i32x4_t rA1 = make_rsrc(g.c1 ? g.a1 : g.a0, rowsA * (g.c1 ? g.lda1 : g.lda0) * SZB);
  const i32x4_t rA2 = make_rsrc(g.c2 ? g.a2 : g.a0, rowsA * (g.c2 ? g.lda2 : g.lda0) * SZB);
  auto issue_a = [&](int ch, int slot) __attribute__((always_inline)) {      // (every branch wave-uniform)
    if (!loader) return;
    const unsigned base = lds0 + slot * TS_ASLOT + wave * 1024;
    if (ch >= ncm) {
      const unsigned so = (unsigned)((ch - ncm) * BKE) * SZB;
#pragma unroll
      for (int j = 0; j < LA; ++j) blds16(rA2, off2[j], so, base + j * PASSB);
    } else if (ch * BKE < g.c0) {
      const unsigned so = (unsigned)(ch * BKE) * SZB;
#pragma unroll
      for (int j = 0; j < LA; ++j) blds16(rA0, off0[j], so, base + j * PASSB);
    } else {
      const unsigned so = (unsigned)(ch * BKE - g.c0) * SZB;
#pragma unroll
      for (int j = 0; j < LA; ++j) blds16(rA1, off1[j], so, base + j * PASSB);
    }
  };

  using Gnl = GnInloop<TM, (NL - NLD) * 64 ? (NL - NLD) * 64 : 64>;
  Gnl gnl;
  typename Gnl::XSet xs;
  const bool gnl_raw = GNP == 2 && g.gnp_raw != nullptr && tn == 0;     // the first column tile also writes the un-normalised operand copy
  if constexpr (GNP == 2) {
    auto real_below = [&](int q) __attribute__((always_inline)) { const int b = q / P; return b * T + min(q - b * P, T); };
    const int rlo = real_below(max(q0 - 1, 0)), rhi = real_below(min(q0 + TS_BM - 1, MP));
    gnl.setup(g, q0, producer ? tid - NLD * 64 : 0, rlo, rhi);
    if (producer) gnl.load(g, 0, xs);                            // chunk 0's rows fly while the statistics are turned into (mean, rstd)
    gnl.table(g, tid, tabmem);
  }
  if constexpr (GNP == 1 || GNP == 3) {
    if (g.gnp_x != nullptr) {
      // the real rows behind padded rows [q0 - 1, q0 + 127): f(q) = number of real rows with a padded index below q
      auto real_below = [&](int q) __attribute__((always_inline)) { const int b = q / P; return b * T + min(q - b * P, T); };
      const int rlo = real_below(max(q0 - 1, 0)), rhi = real_below(min(q0 + TS_BM - 1, MP));
      GnPrologue<TM, NS2VC_TS_GNP_XB, GNP == 3> gpro;
      gpro.begin(g, rlo, rhi, tm, tid, 64 * NW, coop ? tn : 0, coop ? nb_n : 1);
      gpro.finish(g, tid, aring);                                // (its table lives in the activation ring: nothing has been issued into it yet)
    }
  }
  TS_STAMP(7);
  if constexpr (GNP == 2) {
    // chunk 0 is produced here (ncm >= 1), chunk 1's rows are requested; a single-tap chunk 1 (fused 1x1 segment behind ONE main chunk) comes by DMA
    if (producer) {
      gnl.produce(g, 0, xs, aring, tabmem, gnl_raw);
      if (1 < ncm) gnl.load(g, 1, xs);
    }
    if (1 >= ncm && 1 < NCH) issue_a(1, 1);
  } else {
    issue_a(0, 0);
    if (NCH > 1) issue_a(1, 1);
  }

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, hi = lane >> 5;
  const int swb = (l31 >> 1) & 7;
  TS_STAMP(1);
  // ---- steps.  The two roles run their own lean loops (r5 session 2: one shared loop made every wave walk the issue cursors, the
  // source branches and the counted-wait chain -- some 250 mostly scalar instructions and two dozen branches per step -- in front of
  // 256 cycles of MFMA); they meet at one s_barrier per step.
  if (consumer) {
    const char* const arow = aring + (wm * WM + l31) * TS_ROW;
    const char* const brow = wring + (wn * 64 + l31) * TS_ROW;
    unsigned long long t_bar = 0, t_mma = 0, t0 = 0;
    (void)t_bar; (void)t_mma; (void)t0;
    auto step = [&](auto TAU, int aoff, int woff, bool meet = true) __attribute__((always_inline)) {
      constexpr int tau = decltype(TAU)::value;
      const int swa = ((l31 + tau) >> 1) & 7;
      const char* ap = arow + aoff + tau * TS_ROW;
      const char* bp = brow + woff;
      TS_CLK(t0);
      if (meet) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my fragment reads of the slots about to be refilled are done
        __builtin_amdgcn_s_barrier();
      }
      TS_ACC(t_bar, t0);
      // every fragment read of the step first (one consumer wave per SIMD: nothing else hides the LDS round trip); the compiler's
      // counted waits then release the MFMAs one k-slab at a time
      if (NS2VC_TS_ABLATE & 1) return;
      u32x4_t af[NKK][MT], bf[NKK][NT];
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        const int ks = KS ? 2 * kh + kk : kk;
        const int coffa = ((2 * ks + hi) ^ swa) * 16, coffb = ((2 * ks + hi) ^ swb) * 16;
#pragma unroll
        for (int i = 0; i < MT; ++i) af[kk][i] = *reinterpret_cast<const u32x4_t*>(ap + i * 32 * TS_ROW + coffa);
#pragma unroll
        for (int j = 0; j < NT; ++j) bf[kk][j] = *reinterpret_cast<const u32x4_t*>(bp + j * 32 * TS_ROW + coffb);
      }
#if NS2VC_CONS_PF
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (NS2VC_TS_ABLATE & 4) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
#pragma unroll
          for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(af[kk][i]));
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(bf[kk][j]));
        }
        return;
      }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) MmaT<TM>::mma(acc[i][j], af[kk][i], bf[kk][j]);
      TS_ACC(t_mma, t0);
    };
    int aoff = 0, woff = 0;
    auto nextw = [&]() __attribute__((always_inline)) { woff = woff == (SW - 1) * WSLOT ? 0 : woff + WSLOT; };
    for (int ch = 0; ch < ncm; ++ch) {
      step(std::integral_constant<int, 0>{}, aoff, woff); nextw();
      step(std::integral_constant<int, 1>{}, aoff, woff, !NS2VC_TS_CHUNK); nextw();     // (chunk-granular: the chunk's taps back to back)
      step(std::integral_constant<int, 2>{}, aoff, woff, !NS2VC_TS_CHUNK); nextw();
      aoff = aoff == 2 * TS_ASLOT ? 0 : aoff + TS_ASLOT;
    }
    for (int ch = 0; ch < ncs; ++ch) {
      step(std::integral_constant<int, 1>{}, aoff, woff); nextw();
      aoff = aoff == 2 * TS_ASLOT ? 0 : aoff + TS_ASLOT;
    }
    TS_PUT(8, t_bar); TS_PUT(9, t_mma);
  } else {
#if NS2VC_TS_CHUNK
    // One wait + one barrier per chunk.  Needed at chunk ch: its weight tiles and its rows.  Loads return in order, so what may stay in flight is what this
    // wave issued after the later of the two: DG = 2: the tiles and rows of chunk ch + 1 (issued at step ch - 1; at ch = 0 only the rows of chunk 1 follow
    // those of chunk 0 in the prologue); DG = 1: the rows of chunk ch + 1 (issued behind the tiles of chunk ch at step ch - 1).
    unsigned long long t_wait = 0, t_bar = 0, t_iss = 0, t0 = 0;
    (void)t_wait; (void)t_bar; (void)t_iss; (void)t0;
    int aslot = 2;                                               // ring slot of the rows issued next (chunk ch + 2)
    if constexpr (GNP == 2) {
      if (producer) {
        // producer waves: one chunk ahead of the consumers.  After barrier ch the consumers read chunk ch; chunk ch + 1 is built from the registers
        // requested a step ago into the slot chunk ch - 2 left, then chunk ch + 2's rows are requested.  (All waits here are the compiler's.)
        for (int ch = 0; ch < NCH; ++ch) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the rows of chunk ch (written during step ch - 1) are in LDS
          __builtin_amdgcn_s_barrier();
          const int c1 = ch + 1;
          if (c1 < ncm) {
            gnl.produce(g, c1, xs, aring + (c1 % 3) * TS_ASLOT, tabmem, gnl_raw);
            if (c1 + 1 < ncm) gnl.load(g, c1 + 1, xs);
          }
        }
      } else {
        // DMA waves: the tiles of chunk ch + DG, and the rows of a single-tap chunk ch + 2.  Behind the tiles (and DMA rows) of chunk ch: DG = 2: the tiles
        // and DMA rows of chunk ch + 1 (step ch - 1, or the prologue); DG = 1: the DMA rows of chunk ch + 1.
        auto nW = [&](int c) __attribute__((always_inline)) { return c < NCH ? (c < ncm ? 3 * LB : LB) : 0; };
        auto nA = [&](int c) __attribute__((always_inline)) { return (c >= ncm && c < NCH) ? LA : 0; };
        for (int ch = 0; ch < NCH; ++ch) {
          const int allow = (DG == 2 ? nW(ch + 1) : 0) + nA(ch + 1);
          TS_CLK(t0);
          TsWait<0, 3 * LB + LA>::run(__builtin_amdgcn_readfirstlane(allow));
          TS_ACC(t_wait, t0);
          __builtin_amdgcn_s_barrier();
          TS_ACC(t_bar, t0);
          if (!(NS2VC_TS_ABLATE & 2)) {
            if (ch + DG < NCH) issue_wchunk(ch + DG);
            if (ch + 2 >= ncm && ch + 2 < NCH) issue_a(ch + 2, aslot);
          }
          aslot = aslot == 2 ? 0 : aslot + 1;
          TS_ACC(t_iss, t0);
        }
      }
    } else if (!loader) {
      for (int ch = 0; ch < NCH; ++ch) __builtin_amdgcn_s_barrier();      // (prologue / epilogue waves)
    } else {
    for (int ch = 0; ch < NCH; ++ch) {
      TS_CLK(t0);
      if (ch + 1 >= NCH) wait_vmcnt<0>();
      else if (DG == 1 || ch == 0) wait_vmcnt<LA>();
      else if (ch + 1 < ncm) wait_vmcnt<3 * LB + LA>();
      else wait_vmcnt<LB + LA>();
      TS_ACC(t_wait, t0);
      __builtin_amdgcn_s_barrier();
      TS_ACC(t_bar, t0);
      if (!(NS2VC_TS_ABLATE & 2)) {
        if (ch + DG < NCH) issue_wchunk(ch + DG);
        if (ch + 2 < NCH) issue_a(ch + 2, aslot);
      }
      aslot = aslot == 2 ? 0 : aslot + 1;
      TS_ACC(t_iss, t0);
    }
    }
#else
    // Invariant: at step s everything issued before the weight tile of step s (issued FIRST at step s-D, or above) has landed.  Within a
    // step the weight tile goes before the activation chunk.  So what may still be in flight at step s is the chunk issued behind that
    // tile at step s-D and everything of steps s-D+1 .. s-1: hw / ha hold the piece counts of the last D steps ([0] = step s-1).  A chunk's
    // rows are issued at the first step of the chunk two before it (`sa`; the prologue's second chunk counts as step -1): usually further
    // back than the tile (landed with it), but at the head of the loop and around the single-step chunks of the 1x1 segment only 4 or 2
    // steps back -- then the wait spares only what was issued after them.
    int s = 0, aslot = 0, wslot = 0;
    int hw[D], ha[D];                                            // weight / activation pieces issued at step s-1-i
#pragma unroll
    for (int i = 0; i < D; ++i) { hw[i] = 0; ha[i] = 0; }
    ha[0] = NCH > 1 ? LA : 0;
    int fs_m2 = -1, fs_m1 = -1;                                  // first steps of the chunks two / one before this one (-1: the prologue)
    unsigned long long t_wait = 0, t_bar = 0, t_iss = 0, t0 = 0;
    (void)t_wait; (void)t_bar; (void)t_iss; (void)t0;
    for (int ch = 0; ch < NCH; ++ch) {
      const int ntau = ch < ncm ? 3 : 1;
      const int fs = s;
      for (int ti = 0; ti < ntau; ++ti, ++s) {
        const int lb = (ti == 0 && ch > 0) ? s - fs_m2 : D + 1;    // steps back to the issue of this chunk's rows (if they are needed now)
        int allow = 0;
        if (lb <= D) {
#pragma unroll
          for (int i = 0; i < D - 1; ++i) allow += (i <= lb - 2) ? hw[i] + ha[i] : 0;
        } else {
#pragma unroll
          for (int i = 0; i < D - 1; ++i) allow += hw[i] + ha[i];
          allow += ha[D - 1];
        }
        TS_CLK(t0);
        TsWait<0, D * LB + 3 * LA>::run(allow);
        TS_ACC(t_wait, t0);
        __builtin_amdgcn_s_barrier();
        TS_ACC(t_bar, t0);
#pragma unroll
        for (int i = D - 1; i > 0; --i) { hw[i] = hw[i - 1]; ha[i] = ha[i - 1]; }
        hw[0] = 0; ha[0] = 0;
        if (s + D < S && !(NS2VC_TS_ABLATE & 2)) {               // weight tile s+D -> the slot tile s-1 just left
          issue_w(wslot == 0 ? SW - 1 : wslot - 1);
          hw[0] = LB;
        }
        if (ti == 0 && ch + 2 < NCH && !(NS2VC_TS_ABLATE & 2)) { // chunk ch+2 -> the slot chunk ch-1 just left
          issue_a(ch + 2, aslot == 0 ? 2 : aslot - 1);
          ha[0] = LA;
        }
        TS_ACC(t_iss, t0);
        if (++wslot == SW) wslot = 0;
      }
      if (++aslot == 3) aslot = 0;
      fs_m2 = fs_m1; fs_m1 = fs;
    }
#endif
    TS_PUT(4, t_wait); TS_PUT(5, t_bar); TS_PUT(6, t_iss);
  }
  TS_STAMP(2);

  if (ewave < 0) return;                                         // loaders beyond the epilogue's eight waves (s_barrier counts live waves only)
  // ---- epilogue: per 32-row slab the four consumer waves stage their 32 x 64 tile in LDS (re-using the rings), then each of the
  // eight waves moves 16 whole rows out (16-B fp32 / 8-B 16-bit stores, coalesced); bias, residual, statistics; padded -> real rows
  const int kg = ewave >> 2, wq = ewave & 3;                     // row half inside a slab; slab slot (KS: 32-row slab of the tile, else wave tile)
  // The eight epilogue waves each own rows kg*16 .. +15 of one staged 32 x 64 slab `wq`.  Plain layout: slab wq = consumer wq's current 32-row
  // block (MT rounds); K-split layout: all four 32-row slabs of the tile are staged at once, each by the two consumers (K halves) of its
  // row half, and the readers add the pair.
  const int em = KS ? 0 : wq / WGN, en = KS ? 0 : wq % WGN;
  constexpr int EP = 64 + 4, SLAB = 32 * EP;
  float* const etw = reinterpret_cast<float*>(smem);             // staging: [K half][4 slabs] (KS) / [4 slabs]
  float* of = g.out_f32;
  TM* oo = reinterpret_cast<TM*>(g.out_op);
  constexpr int LPR = 16, RPI = 4, NIT = 4;
  const int rsub = lane / LPR, cq = lane % LPR;
  const int ncol = n0 + en * 64 + cq * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + ncol);
  constexpr int ROUNDS = KS ? 1 : MT;
  const int qw0 = q0 + (KS ? wq * 32 : em * WM);                 // first padded row of what this wave's statistics cover
  constexpr int QSPAN = KS ? 32 : WM;
  const int b0 = min(qw0 / P, g.B - 1);                          // its batch item; T >= 66 > QSPAN: the rows touch b0 and at most b0 + 1
  float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
#pragma unroll
  for (int mt = 0; mt < ROUNDS; ++mt) {
    lds_barrier();                                               // rings (or the previous slab) are free
    if (kg == 1) {                                               // (the consumer waves)
      if constexpr (KS) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          float* const e = etw + (kh * 4 + wm * 2 + i) * SLAB;
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) e[(8 * (r >> 2) + 4 * hi + (r & 3)) * EP + j * 32 + l31] = acc[i][j][r];
        }
      } else {
        float* const e = etw + wq * SLAB;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) e[(8 * (r >> 2) + 4 * hi + (r & 3)) * EP + j * 32 + l31] = acc[mt][j][r];
      }
    }
    lds_barrier();
    int mrow[NIT];
    bool okr[NIT], first[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int rl = kg * 16 + k * RPI + rsub;                   // row inside the slab
      const int rt = KS ? wq * 32 + rl : em * WM + mt * 32 + rl; // row inside the tile
      const int q = q0 + rt;
      const int b = q / P, t = q - b * P;
      okr[k] = rt < TS_BMO && q < MP && t < T;
      mrow[k] = okr[k] ? q - b : 0;
      first[k] = b == b0;
    }
    float4 rr[NIT];
    if (g.res) {                                                 // residual rows first (res may alias out_f32 element-for-element)
#pragma unroll
      for (int k = 0; k < NIT; ++k) rr[k] = *reinterpret_cast<const float4*>(g.res + (size_t)mrow[k] * g.ldres + ncol);
    } else {
#pragma unroll
      for (int k = 0; k < NIT; ++k) rr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 vv[NIT];
    const float* const er = etw + wq * SLAB;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int row = kg * 16 + k * RPI + rsub;
      float4 a = *reinterpret_cast<const float4*>(er + row * EP + cq * 4);
      if constexpr (KS) {                                        // K half 0 + K half 1, always in this order
        const float4 a1 = *reinterpret_cast<const float4*>(er + 4 * SLAB + row * EP + cq * 4);
        a.x += a1.x; a.y += a1.y; a.z += a1.z; a.w += a1.w;
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (okr[k]) {
        v.x = a.x + bv.x + rr[k].x; v.y = a.y + bv.y + rr[k].y; v.z = a.z + bv.z + rr[k].z; v.w = a.w + bv.w + rr[k].w;
        const float ps = (v.x + v.y) + (v.z + v.w), pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        if (first[k]) { gs0 += ps; gq0 += pq; } else { gs1 += ps; gq1 += pq; }
      }
      vv[k] = v;
    }
    if constexpr (SOL) {
     if (g.sol_coef) {
      // r6 (tested option, engine switch fuse_solver): the sampling loop's solver update on the tile this workgroup holds (conv_out: every latent row exactly
      // once): the result is x0, the state rows are four more row loads -- one launch, 73 MB of reads and 15 MB of writes less per step than the separate
      // kernel.  Measured time-neutral to 0.4 % slower (profiles/r06_ab_fuse_solver.txt): the separate kernel streams at HBM speed, these loads sit in the
      // epilogue's dependent chain (hoisting them above the staging and pre-touching the lines from the loader waves moved nothing).
      const SolverCoef sk = solver_coef(g.sol_coef + (size_t)(*g.sol_step) * g.sol_ncoef);
      float4 sxe[NIT], sxb[NIT], sd1[NIT], smp[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const size_t o = (size_t)mrow[k] * g.sol_ld + ncol;
        sxe[k] = *reinterpret_cast<const float4*>(g.sol_xe + o); sxb[k] = *reinterpret_cast<const float4*>(g.sol_xbar + o);
        sd1[k] = *reinterpret_cast<const float4*>(g.sol_d1 + o); smp[k] = *reinterpret_cast<const float4*>(g.sol_mprev + o);
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (okr[k]) {
          const size_t o = (size_t)mrow[k] * g.sol_ld + ncol;
          float4 oxe, oxb, od1, om;
          solver_upd(sk, vv[k].x, sxe[k].x, sxb[k].x, sd1[k].x, smp[k].x, oxe.x, oxb.x, od1.x, om.x);
          solver_upd(sk, vv[k].y, sxe[k].y, sxb[k].y, sd1[k].y, smp[k].y, oxe.y, oxb.y, od1.y, om.y);
          solver_upd(sk, vv[k].z, sxe[k].z, sxb[k].z, sd1[k].z, smp[k].z, oxe.z, oxb.z, od1.z, om.z);
          solver_upd(sk, vv[k].w, sxe[k].w, sxb[k].w, sd1[k].w, smp[k].w, oxe.w, oxb.w, od1.w, om.w);
          out_f4(g.sol_xe + o, oxe.x, oxe.y, oxe.z, oxe.w);
          if (g.sol_op_pair) {                             // hi + lo operand pair, rows of 2 * sol_ld columns
            TM* const q = reinterpret_cast<TM*>(g.sol_xe_op) + o + (size_t)mrow[k] * g.sol_ld;
            out_op4<TM>(q, oxe.x, oxe.y, oxe.z, oxe.w);
            out_op4<TM>(q + g.sol_ld, op_rest<TM>(oxe.x), op_rest<TM>(oxe.y), op_rest<TM>(oxe.z), op_rest<TM>(oxe.w));
          } else {
            out_op4<TM>(reinterpret_cast<TM*>(g.sol_xe_op) + o, oxe.x, oxe.y, oxe.z, oxe.w);
          }
          out_f4(g.sol_xbar + o, oxb.x, oxb.y, oxb.z, oxb.w);
          out_f4(g.sol_d1 + o, od1.x, od1.y, od1.z, od1.w);
          out_f4(g.sol_mprev + o, om.x, om.y, om.z, om.w);
        }
      }
     }
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      if (okr[k]) {
        if (of) out_f4(of + (size_t)mrow[k] * g.ldo_f32 + ncol, vv[k].x, vv[k].y, vv[k].z, vv[k].w);
        if (oo) out_op4<TM>(oo + (size_t)mrow[k] * g.ldo_op + ncol, vv[k].x, vv[k].y, vv[k].z, vv[k].w);
      }
    }
  }
  if (g.stats) {
    // fixed shuffle tree over the lanes that share a 16-channel block (4 column quads x the row lanes), then ONE int64 fixed-point
    // atomic per (batch item, block, moment): order-independent => deterministic
    double d0 = gs0, d1 = gq0, d2 = gs1, d3 = gq1;
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o); }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o); }
    if (rsub == 0 && (cq & 3) == 0 && qw0 < MP) {
      const int blk = ncol >> 4, nblk = g.N >> 4;
      unsigned long long* st = reinterpret_cast<unsigned long long*>(g.stats) + ((size_t)b0 * nblk + blk) * 2;
      atomicAdd(st, (unsigned long long)llrint(d0 * GN_SUM_SCALE));
      atomicAdd(st + 1, (unsigned long long)llrint(d1 * GN_SQ_SCALE));
      if (b0 + 1 < g.B && (b0 + 1) * P < qw0 + QSPAN) {
        atomicAdd(st + 2 * nblk, (unsigned long long)llrint(d2 * GN_SUM_SCALE));
        atomicAdd(st + 2 * nblk + 1, (unsigned long long)llrint(d3 * GN_SQ_SCALE));
      }
    }
  }
#if NS2VC_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (only so that the last stamp includes the store drain)
  if (tr && wave == EOFF && lane == 0) tr[3] = TS_NOW();
#endif
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Tile-major image of a k = 3 conv weight for conv3ts_kernel (GemmArgs.w_tiled).  rows: [N][K] fp32 as ns2vc_pack_weight takes them, K = 3 * ctot + c2
// (k = tap * ctot + c, then the c2 columns of a fused 1x1 segment); N is padded to a multiple of 128 with zero rows.  Output, in the operand type:
// [Np / 64 column groups][S steps][64 rows][128 B], S = 3 * (ctot / bke) + c2 / bke steps in the kernel's consumption order (chunk-major, tap-minor, then the
// single-tap chunks); inside a block byte r * 128 + p * 16 holds the 16-B chunk (p ^ ((r >> 1) & 7)) of row r: the bank-conflict-free LDS image, so the DMA is
// lane-linear.
hipError_t pack_conv3_tiled(const float* rows, int N, int ctot, int c2, int prec, std::vector<unsigned char>& out) {
  const int bke = prec == PREC_F32 ? 32 : 64, esz = prec == PREC_F32 ? 4 : 2, epc = 16 / esz;
  if (N <= 0 || ctot < bke || (ctot % bke) || (c2 % bke)) return hipErrorInvalidValue;
  const int K = 3 * ctot + c2, Np = (N + 127) / 128 * 128;
  const int ncm = ctot / bke, ncs = c2 / bke, S = 3 * ncm + ncs;
  out.assign((size_t)Np * K * esz, 0);
  for (int tg = 0; tg < Np / 64; ++tg)
    for (int s = 0; s < S; ++s) {
      const int koff = s < 3 * ncm ? (s % 3) * ctot + (s / 3) * bke : 3 * ctot + (s - 3 * ncm) * bke;
      unsigned char* blk = out.data() + ((size_t)tg * S + s) * 8192;
      for (int r = 0; r < 64; ++r) {
        const int n = tg * 64 + r;
        if (n >= N) continue;
        for (int p = 0; p < 8; ++p) {
          const int lc = p ^ ((r >> 1) & 7);
          for (int e = 0; e < epc; ++e) {
            const float v = rows[(size_t)n * K + koff + lc * epc + e];
            unsigned char* dst = blk + r * 128 + p * 16 + e * esz;
            if (prec == PREC_F32) memcpy(dst, &v, 4);
            else { const uint16_t q = f32_to_op16_bits(v, prec); memcpy(dst, &q, 2); }
          }
        }
      }
    }
  return hipSuccess;
}

// (+ 4 KB behind the rings: the (mean, rstd) table of the in-loop GroupNorm, which has to outlive the prologue)
static constexpr size_t ts_lds_bytes(int bn) { return (size_t)3 * TS_ASLOT + (size_t)(bn == 64 ? TsRing<64>::SW : TsRing<128>::SW) * bn * TS_ROW + 4096; }

bool convts_eligible(const GemmArgs& g, int prec) {
  const int bke = prec == PREC_F32 ? 32 : 64;
  if (g.taps != 3 || g.tmode != TMODE_SAME || g.Tin != g.Tout || g.Tin < 66 || g.geglu || g.rowstats || g.ln_stats) return false;
  if ((g.N % 64) || (g.c0 % bke) || (g.c1 % bke) || (g.c2 % bke) || g.c0 + g.c1 < bke) return false;
  if ((unsigned long long)g.B * (g.Tin + 1) > 0x7fff0000ull) return false;
  if (g.sol_coef && (g.N != 128 || !g.sol_step || !g.sol_xe || !g.sol_xe_op || !g.sol_xbar || !g.sol_d1 || !g.sol_mprev || (g.sol_ld & 3))) return false;
  return true;
}
// BN: 128-column tiles where they still give the chip one round of workgroups, 64 otherwise (and for N that is no multiple of 128)
static int g_bn128_min = 160;
void set_convts_bn128_min(int wgs) { g_bn128_min = wgs > 0 ? wgs : 160; }
int convts_bn_for(const GemmArgs& g, int bn128_min) {
  const long long nbm = ((long long)g.B * (g.Tin + 1) + TS_BMO - 1) / TS_BMO;
  return ((g.N % 128) == 0 && nbm * (g.N / 128) >= bn128_min) ? 128 : 64;
}
int convts_default_bn(const GemmArgs& g) { return (g.conv_bn == 64 || (g.conv_bn == 128 && g.N % 128 == 0)) ? g.conv_bn : convts_bn_for(g, g_bn128_min); }
int convts_row_blocks(const GemmArgs& g) { return (int)(((long long)g.B * (g.Tin + 1) + TS_BMO - 1) / TS_BMO); }

template <typename TM, int BN, int NL, bool KS> static hipError_t launch_ts_cfg(const GemmArgs& g, hipStream_t s) {
  const int nbm = convts_row_blocks(g), nbn = g.N / BN;
  int nb = nbm * nbn;
  if (g.gnp_x && g.gnp_sync && nbn > 1) nb = 8 * ((nbm + 7) / 8) * nbn;      // cooperative prologue: row blocks per XCD, padded
  // GroupNorm in front: inside the loop (GnInloop: eight non-consumer waves = four DMA + four producer waves, chunk-granular loop, plain consumer layout) unless
  // the caller asks for the materialising prologue (algo == 2)
  constexpr bool HAS_INLOOP = NS2VC_TS_CHUNK && NL == 8 && !KS;
  if (g.sol_coef) {                                               // the solver-update epilogue: its own instantiations (one 128-column tile, eight loaders)
    constexpr bool HAS_SOL = BN == 128 && NL == 8 && !KS;
    if constexpr (HAS_SOL) {
      if (g.gnp_x && g.gnp_pair) {
        if constexpr (!std::is_same<TM, float>::value) hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 3, KS, true>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
        else return hipErrorInvalidValue;
      }
      else if (g.gnp_x) hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 1, KS, true>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);     // (always the materialising prologue)
      else hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 0, KS, true>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
    } else return hipErrorInvalidValue;
  }
  else if (g.gnp_x && g.gnp_pair) {                                    // a hi + lo operand pair is something the materialising prologue writes (its own instantiation)
    if constexpr (!std::is_same<TM, float>::value && NL == 8 && !KS) hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 3, KS>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
    else return hipErrorInvalidValue;
  }
  else if (g.gnp_x && HAS_INLOOP && g.algo != 2) {
    if constexpr (HAS_INLOOP) hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 2, KS>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
  }
  else if (g.gnp_x) hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 1, KS>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
  else hipLaunchKernelGGL((conv3ts_kernel<TM, BN, NL, 0, KS>), dim3(nb), dim3(64 * (NL + 4)), ts_lds_bytes(BN), s, g);
  return hipGetLastError();
}
template <typename TM> static hipError_t launch_ts_typed(const GemmArgs& g, int bn, int nl, int ks, hipStream_t s) {
  if (bn == 64 && nl == 4) return ks ? launch_ts_cfg<TM, 64, 4, true>(g, s) : launch_ts_cfg<TM, 64, 4, false>(g, s);
  if (bn == 64 && nl == 8) return ks ? launch_ts_cfg<TM, 64, 8, true>(g, s) : launch_ts_cfg<TM, 64, 8, false>(g, s);
  if (bn == 128 && nl == 4) return launch_ts_cfg<TM, 128, 4, false>(g, s);
  if (bn == 128 && nl == 8) return launch_ts_cfg<TM, 128, 8, false>(g, s);
  return hipErrorInvalidValue;
}
// bn 64 | 128 (0: heuristic), nl 4 | 8 loader waves (0: default), ks: the K-split consumer layout of the 64-column tile (-1: default)
hipError_t launch_convts(const GemmArgs& g, int prec, int bn, int nl, int ks, hipStream_t s) {
  if (!convts_eligible(g, prec)) return hipErrorInvalidValue;
  if (!bn) bn = convts_default_bn(g);
  if (!nl) nl = NS2VC_TS_NL_DEFAULT;
  if (ks < 0) ks = NS2VC_TS_KS_DEFAULT;
  if (g.sol_coef) { bn = 128; nl = 8; ks = 0; }                   // (the one configuration the solver epilogue is instantiated for)
  if (g.N % bn) return hipErrorInvalidValue;
  switch (prec) {
    case PREC_BF16: return launch_ts_typed<bf16_t>(g, bn, nl, ks, s);
    case PREC_F16: return launch_ts_typed<f16_t>(g, bn, nl, ks, s);
    case PREC_F32: return launch_ts_typed<float>(g, bn, nl, ks, s);
    default: return hipErrorInvalidValue;
  }
}

template <typename K> static hipError_t ts_set_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
template <typename TM> static hipError_t ts_init_typed() {
  hipError_t e = hipSuccess;
#define NS2VC_TS_SET(BN_, NL_)                                                                      \
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, BN_, NL_, 0>, ts_lds_bytes(BN_));     \
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, BN_, NL_, 1>, ts_lds_bytes(BN_))
  NS2VC_TS_SET(64, 4); NS2VC_TS_SET(128, 4); NS2VC_TS_SET(64, 8); NS2VC_TS_SET(128, 8);
#if NS2VC_TS_CHUNK
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 8, 2>, ts_lds_bytes(64));
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 128, 8, 2>, ts_lds_bytes(128));
#endif
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 128, 8, 0, false, true>, ts_lds_bytes(128));
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 128, 8, 1, false, true>, ts_lds_bytes(128));
  if constexpr (!std::is_same<TM, float>::value) {
    if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 128, 8, 3, false, true>, ts_lds_bytes(128));
    if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 8, 3>, ts_lds_bytes(64));
    if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 128, 8, 3>, ts_lds_bytes(128));
  }
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 4, 0, true>, ts_lds_bytes(64));
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 4, 1, true>, ts_lds_bytes(64));
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 8, 0, true>, ts_lds_bytes(64));
  if (e == hipSuccess) e = ts_set_lds(conv3ts_kernel<TM, 64, 8, 1, true>, ts_lds_bytes(64));
#undef NS2VC_TS_SET
  return e;
}
hipError_t init_convts_attributes() {
  hipError_t e = ts_init_typed<float>();
  if (e == hipSuccess) e = ts_init_typed<bf16_t>();
  if (e == hipSuccess) e = ts_init_typed<f16_t>();
  return e;
}

}  // namespace ns2vc
