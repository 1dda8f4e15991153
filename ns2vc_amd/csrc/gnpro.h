// GroupNorm-apply prologue shared by the implicit-GEMM kernels (gemm.hip, convts.hip), gfx950.
#pragma once
#include "common.h"
#include <type_traits>

namespace ns2vc {

// ---------------------------------------------------------------------------
// GroupNorm-apply prologue (built in r3, shipped in r4; GemmArgs.gnp_*).  The step had 51 gn_apply launches whose only job is to turn fp32 rows into
// the operand rows ONE following GEMM reads.  A consumer-side fusion that normalises while it loads (r1 conv3gn) pays the
// normalisation once per tap and column tile on the MFMA waves' critical path; a producer-side fusion needs the whole-item
// statistic (r2 gn_producer, r3 convgn).  This one keeps the K loop and its LDS-DMA untouched: every workgroup first
// materialises the operand rows ITS tile will read -- its BM output rows plus one halo row either side for k = 3, all c0
// channels -- in the operand tensor a0, with exactly gn_apply_kernel's arithmetic (bit-identical rows), waits for its own
// stores, and then runs as before: the DMA reads the rows back from the L2 they were just written through.  Halo rows are
// produced redundantly by the two neighbouring row blocks with identical bytes.  The column tiles of one row block either do the
// same (no ordering between workgroups needed) or -- with GemmArgs.gnp_sync, the engine's default -- build a SHARE of the block's
// rows each and wait for the others' behind an arrival count in the L2 of the XCD they all run on (`finish`; r4, -1 .. -2 % of the
// step: the redundant form spent 3x / 4x the SiLU work at 384 / 512 channels).  The input may be the channel concat of two tensors
// with their own statistics (gnp_x1), and the un-normalised operand copy a 1x1 shortcut reads later can be written along (gnp_raw):
// with those, every GroupNorm of the bench plan is a prologue (r3: 51 gn_apply launches, r4: none).
// ---------------------------------------------------------------------------
#ifndef NS2VC_GNP_WT
#define NS2VC_GNP_WT 1
#endif
// In two halves: `begin` issues EVERY load of the prologue -- the first batch of fp32 rows, the int64 statistics of the (item, group)
// pairs this tile touches, gamma / beta and the time scale / shift rows --, `finish` does the arithmetic and the stores.  (They run back to
// back: hoisting `begin` above the kernel's row-offset set-up was measured and lost, see NS2VC_GNP_SPLIT.)
// PAIR: the rows are written as a hi + lo operand pair (GemmArgs.gnp_pair; its own instantiation -- as a run-time test in the row loop it cost EVERY
// prologue launch 1.5 - 1.8 %, through seven more spilled registers)
template <typename TM, int XB_, bool PAIR = false> struct GnPrologue {
#ifndef NS2VC_GNP_XB
#define NS2VC_GNP_XB 6
#endif
#ifndef NS2VC_GNP_SPIN
#define NS2VC_GNP_SPIN 256           // polls (~0.5 us each) before a workgroup stops waiting for its siblings and builds every row itself
#endif
  static constexpr int XB = XB_;                                            // rows in flight per thread (1: no gain in the loop, 6: -1 %); more only where the tile is alone on its CU anyway
  static constexpr int OFF_BSUM = 256, OFF_OK = 3584;                       // table area (the ring stage nobody has been issued into yet): (mean, rstd) pairs | block sums | flag
  int rlo, rhi, olo, ohi, lim, rln, b_lo, nbi, rl, c, cq, gg, Cg, nshare_, cur;
  unsigned long long* cnt_;
  bool active;
  float4 ga, be, t1, t2, xb[XB];                                            // t1 / t2: the time (scale | shift) quad of item `cur`
  const float* xsrc;                                                        // this thread's column quad in its source tensor (the input may be a concat of two)
  int xld;
  long long sv[2];                                                          // (sum, sum of squares) of ONE 16-channel block of one item (thread = item x block)
  float a[4], b[4];                                                         // y = x * a + b for this quad, item `cur`

  __device__ __forceinline__ void fetch(const GemmArgs& g, int rb) {        // (every lane loads, from a clamped row: a straight-line batch of plain loads)
    (void)g;
#pragma unroll
    for (int k = 0; k < XB; ++k) {
      const int r = max(min(rb + k * rl, lim - 1), rlo);                    // (an empty share still loads a valid row)
      xb[k] = *reinterpret_cast<const float4*>(xsrc + (size_t)r * xld);
    }
  }
  __device__ __forceinline__ int item_of(const GemmArgs& g, int r) const {
    return min((r >= (b_lo + 1) * g.Tin ? 1 : 0) + (r >= (b_lo + 2) * g.Tin ? 1 : 0), nbi - 1);
  }
  __device__ __forceinline__ void load_temb(const GemmArgs& g, int bi) {
    t1 = t2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.gnp_temb) {
      const int C = PAIR ? g.c0 >> 1 : g.c0;
      const float* tp = g.gnp_temb + (size_t)(b_lo + bi) * g.gnp_ldtemb + cq;
      if ((reinterpret_cast<uintptr_t>(tp) & 15) == 0 && (C & 3) == 0) {
        t1 = *reinterpret_cast<const float4*>(tp);
        t2 = *reinterpret_cast<const float4*>(tp + C);
      } else {
        t1 = make_float4(tp[0], tp[1], tp[2], tp[3]);
        t2 = make_float4(tp[C], tp[C + 1], tp[C + 2], tp[C + 3]);
      }
    }
  }
  // (scale, shift) of item `cur` for this thread's quad, from the table in LDS and the vectors in registers
  __device__ __forceinline__ void affine(const GemmArgs& g, const char* smem) {
    float2 mr = reinterpret_cast<const float2*>(smem)[cur * 8 + gg];
    // gfx950 hazard guard (r4, profiles/r04_gn_prologue_rootcause.txt).  Left to the compiler this spot became
    //   ds_read_b64 x3 ; s_waitcnt vmcnt(1) lgkmcnt(2) ; v_pk_mul_f32 v[..], gamma.xy, v[mean:rstd] op_sel:[0,1]
    // and, in kernels running beside the LDS-DMA traffic of the loader waves, the packed product came back as 0.0 in its LOW half
    // for lanes 48-63 of a few waves per launch (inputs verified intact by a scalar recompute of the same registers): the
    // "gamma reads zero" non-determinism of round 3.  With the (mean, rstd) pair landed before the first packed product the
    // launch is bit-reproducible (gnp_probe 10 / 10, 10 captured / eager loops, 12 forwards at the bench shape).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(mr.x), "+v"(mr.y));
    const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
    const float ts[4] = {t1.x, t1.y, t1.z, t1.w}, tf[4] = {t2.x, t2.y, t2.z, t2.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = mr.y * gam[e];
      b[e] = bet[e] - mr.x * a[e];
      if (g.gnp_temb) {
        const float s1 = 1.0f + ts[e];
        a[e] *= s1;
        b[e] = b[e] * s1 + tf[e];
      }
    }
  }
  // rows [rlo_, rhi_) of the flattened (item, frame) index are what the tile reads; `blk` names the tile's row block (its arrival count)
  __device__ __forceinline__ void begin(const GemmArgs& g, int rlo_, int rhi_, int blk, int tid, int nth, int share, int nshare) {
    const int C = PAIR ? g.c0 >> 1 : g.c0, T = g.Tin;
    Cg = C / g.gnp_G;
    rlo = rlo_; rhi = rhi_;
    b_lo = rlo / T;
    nbi = (rhi - 1) / T - b_lo + 1;                                         // <= 3 (the launcher checks T against the tile)
    const int nq = C >> 2;                                                  // float4 quads per row
    rl = nth / nq;                                                          // rows per pass
    const int quad = tid % nq;
    rln = tid / nq;
    c = quad * 4;
    active = rln < rl;
    cq = active ? c : 0;
    gg = cq / Cg;
    // cooperative form (gnp_sync): the nshare workgroups that share these rows (the column tiles of one row block, neighbours on one
    // XCD) build a contiguous share each; [olo, ohi) is mine
    nshare_ = nshare;
    cnt_ = nshare > 1 ? reinterpret_cast<unsigned long long*>(g.gnp_sync) + blk : nullptr;
    const int per = (rhi - rlo + nshare - 1) / nshare;
    olo = min(rlo + share * per, rhi); ohi = min(olo + per, rhi);
    lim = ohi;
    const int c0a = C - g.gnp_c1;                                           // channels of the first source (all of them unless the input is a concat)
    if (cq < c0a) { xsrc = g.gnp_x + cq; xld = g.gnp_ldx; } else { xsrc = g.gnp_x1 + (cq - c0a); xld = g.gnp_ldx1; }
    fetch(g, olo + rln);
    sv[0] = sv[1] = 0;
    const int nblk = C >> 4;
    if (tid < nbi * nblk) {                                                 // <= 3 x 64 threads, one 16-channel block of one item each
      const int bi = tid / nblk, blk = tid - bi * nblk, nblk0 = c0a >> 4;
      const long long* st = blk < nblk0 ? g.gnp_stats + ((size_t)(b_lo + bi) * nblk0 + blk) * 2
                                        : g.gnp_stats1 + ((size_t)(b_lo + bi) * (nblk - nblk0) + (blk - nblk0)) * 2;
      sv[0] = st[0]; sv[1] = st[1];
    }
    ga = *reinterpret_cast<const float4*>(g.gnp_gamma + cq);                // (unconditional loads: inactive threads read column 0)
    be = *reinterpret_cast<const float4*>(g.gnp_beta + cq);
    cur = item_of(g, olo + rln);
    load_temb(g, cur);
  }
  // rows [lo, hi) of this thread's column quad: act(x * a + b) -> operand type, written through to L2.  A thread's rows ascend, so the
  // item they belong to changes at most twice: its (scale, shift) quad is kept and rebuilt behind a branch that is almost never taken
  // (r4: per-row selects among three items' quads were 16 v_cndmask per quad of a VALU-heavy loop, and 24 registers; tools/gnp_trace.py)
  __device__ __forceinline__ void rows(const GemmArgs& g, const char* smem, int lo, int hi, bool fetched) {
    const int T = g.Tin;
    const int rs = lo + rln;
    lim = hi;
    if (!fetched) fetch(g, rs);
    const int it = item_of(g, rs);
    if (it != cur) { cur = it; load_temb(g, cur); }
    affine(g, smem);
    int nxt = (b_lo + cur + 1) * T;
    TM* const dst = reinterpret_cast<TM*>(const_cast<void*>(g.a0));
    TM* const raw = reinterpret_cast<TM*>(g.gnp_raw);
    for (int rb = rs; rb < hi; rb += XB * rl) {
      float4 w[XB];
#pragma unroll
      for (int k = 0; k < XB; ++k) w[k] = xb[k];
      fetch(g, rb + XB * rl);                                               // next batch before this one is stored (clamped: the last one is a dummy)
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        const int r = rb + k * rl;
        if (r < hi) {
          if (r >= nxt) {                                                   // (rows per pass <= 16 < T: never more than one item further)
            ++cur; nxt += T;
            load_temb(g, cur);
            affine(g, smem);
          }
          float y0 = w[k].x * a[0] + b[0], y1 = w[k].y * a[1] + b[1], y2 = w[k].z * a[2] + b[2], y3 = w[k].w * a[3] + b[3];
          if (g.gnp_silu) { y0 = silu_f(y0); y1 = silu_f(y1); y2 = silu_f(y2); y3 = silu_f(y3); }
#if NS2VC_GNP_WT
          out_op4<TM>(dst + (size_t)r * g.lda0 + c, y0, y1, y2, y3);
          if (PAIR) out_op4<TM>(dst + (size_t)r * g.lda0 + (g.c0 >> 1) + c, op_rest<TM>(y0), op_rest<TM>(y1), op_rest<TM>(y2), op_rest<TM>(y3));   // the lo plane of a hi + lo pair
#else
          store_op4<TM>(dst + (size_t)r * g.lda0 + c, y0, y1, y2, y3);
          if (PAIR) store_op4<TM>(dst + (size_t)r * g.lda0 + (g.c0 >> 1) + c, op_rest<TM>(y0), op_rest<TM>(y1), op_rest<TM>(y2), op_rest<TM>(y3));
#endif
          if (raw) out_op4<TM>(raw + (size_t)r * g.lda0 + c, w[k].x, w[k].y, w[k].z, w[k].w);   // the un-normalised operand copy a later 1x1 shortcut reads
        }
      }
    }
  }
  __device__ __forceinline__ void finish(const GemmArgs& g, int tid, char* smem) {
    const int T = g.Tin, G = g.gnp_G;
    float2* const gtab = reinterpret_cast<float2*>(smem);                   // (mean, rstd) of (item - b_lo, group): <= 3 x 8
    double2* const bsum = reinterpret_cast<double2*>(smem + OFF_BSUM);      // per (item - b_lo, 16-channel block): the scaled sums, exact in double
    const int nblk = (PAIR ? g.c0 >> 1 : g.c0) >> 4;
    if (tid < nbi * nblk) bsum[tid] = make_double2((double)sv[0] * (1.0 / GN_SUM_SCALE), (double)sv[1] * (1.0 / GN_SQ_SCALE));
    __syncthreads();
    if (tid < nbi * G) {                                                    // same finalisation as gn_apply_kernel (misc.hip); the sums are exact, their order is free
      const int bi = tid / G, gq = tid - bi * G;
      const int nb = Cg >> 4;
      double ds = 0.0, dq = 0.0;
      for (int j = 0; j < nb; ++j) { const double2 e = bsum[bi * nblk + gq * nb + j]; ds += e.x; dq += e.y; }
      const float inv_nf = 1.0f / ((float)T * (float)Cg);
      const double inv_n = (double)inv_nf * (2.0 - (double)inv_nf * ((double)T * (double)Cg));
      const double mean = ds * inv_n;
      double var = dq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float ve = (float)var + g.gnp_eps;
      float r = rsqrtf(ve);
      r = r * (1.5f - 0.5f * ve * r * r);
      gtab[bi * 8 + gq] = make_float2((float)mean, r);
    }
    __syncthreads();
    if (active) rows(g, smem, olo, ohi, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // my rows are in L2 ...
    __syncthreads();                                        // ... and so are everybody else's: the DMA may read them
    if (nshare_ > 1) {
      // Cooperative form: publish my share, wait (bounded) for the others'.  One 64-bit arrival word per row block, ZERO when the launch
      // starts (the engine carves the words from the statistics pool, so the one clear launch of every forward zeroes them; a word is used by
      // one launch per forward).  A sibling that does not show up in time (not resident yet: nothing guarantees co-scheduling) costs a repeat
      // of the whole range by this workgroup; the values are the same whoever writes them, so the result does not depend on which way it went.
      int* const okf = reinterpret_cast<int*>(smem + OFF_OK);
      if (tid == 0) {
        // The siblings exchange rows through ONE XCD's L2 (the cooperative tile order puts workgroup ids that are equal mod 8 side by
        // side, and the dispatcher deals ids round robin over the XCDs): the rows were written through and acknowledged (vmcnt(0)
        // above), and the count is only ever touched by read-modify-writes, which execute in that L2.  (Agent-scope fences / atomics
        // would be correct under any placement, but on this multi-XCD part they cost an L2 write-back and a trip to the memory side per
        // workgroup: measured, the prologue got slower than the redundant form.)
        // r5: that placement is CHECKED, per row block and launch, not assumed.  The dispatcher's round robin does NOT start at XCD 0 for
        // every launch (measured, profiles/r05_placement_probe.txt: the first launch of a process deals id i to XCD i % 8, later ones
        // start at XCD 7), so there is no absolute "XCD of slot i" to compare with; what matters is that the siblings share an XCD.  Every
        // arrival therefore also counts itself in the nibble of ITS XCC id (HW_REG_XCC_ID) inside the same 64-bit word:
        //   bits 0-15 arrivals | bits 16-47 eight 4-bit per-XCC arrival counts | bit 62 "do not wait" (tests)
        // and a workgroup trusts its siblings' rows only when all nshare arrivals it can see carry its own XCC id.  A sibling that runs
        // elsewhere (a CU-masked stream, another partition mode, a dispatcher that deals differently) works on its own L2's copy of the
        // word: nobody ever sees nshare arrivals, everybody builds all its rows itself after the bounded wait.  The engine zeroes the
        // words at the start of every forward (they live in the statistics pool), so whatever such a launch leaves behind is gone
        // before the word's next use; a word that does not start at zero (> nshare arrivals, foreign nibbles) is never trusted.
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        const unsigned long long inc = 1ull | (1ull << (16 + 4 * xcc));
        const unsigned long long want = (unsigned long long)(unsigned)nshare_ | ((unsigned long long)(unsigned)nshare_ << (16 + 4 * xcc));
        unsigned long long old, v, zero = 0ull;
        asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(old) : "v"(cnt_), "v"(inc) : "memory");
        v = old + inc;
        int ok = v == want;
        const unsigned long long mine = 0xffffull | (0xfull << (16 + 4 * xcc));
        // test flag / a word that did not start at zero / arrivals from another XCC: nothing to wait for
        bool hopeless = (old >> 62) != 0 || (v & 0xffffull) > (unsigned)nshare_ || (v & ~mine) != 0;
        for (int it = 0; !ok && !hopeless && it < NS2VC_GNP_SPIN; ++it) {
          asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(cnt_), "v"(zero) : "memory");
          ok = v == want;
          hopeless = (v & 0xffffull) > (unsigned)nshare_ || (v & ~mine) != 0;
          if (!ok && !hopeless) __builtin_amdgcn_s_sleep(2);
        }
        okf[0] = ok;
      }
      __syncthreads();
      const int ok = okf[0];
      if (!ok) {
        if (tid == 0 && g.gnp_alone) atomicAdd(g.gnp_alone, 1u);
        if (active) rows(g, smem, rlo, rhi, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();                                      // (the table area is free from here on)
    }
  }
};

// ---------------------------------------------------------------------------
// r6: the GroupNorm-apply INSIDE the K loop of conv3ts_kernel (GemmArgs.algo == 0 with gnp_x set).
//
// The prologue above materialises a tile's normalised rows in global memory and reads them back by LDS-DMA: a chain of dependent round
// trips (statistics + fp32 rows in, operand rows out and acknowledged, sibling arrival, first DMA) of 11-14 k cycles in front of a K loop
// that at levels 0-1 is only 3-8 k cycles long (profiles/r05_ts_trace_gnp.txt) -- two thirds of the 31 level-0..2 conv launches of a step.
// conv3ts_kernel's loader waves have nothing to do between their DMA issues, and an activation chunk is loaded ONCE per chunk there (not once
// per tap as in the r1 conv3gn kernel, whose normalisation sat on the MFMA waves' critical path): so the loader waves now PRODUCE the chunk --
// fp32 rows global -> VGPR (one step ahead), y = act(x * a + b) with exactly the prologue's / gn_apply_kernel's arithmetic (bit-identical
// operand values), rounded to the operand type and written straight into the ring slot in the swizzled image the consumers read.  Nothing is
// written to or re-read from global memory, no workgroup waits for another, and the first chunk is ready one load latency after the
// statistics.  The column tiles of a row block each build the rows themselves (from the L2 they share); the redundant SiLU work that made
// that form lose as a PROLOGUE (r4) runs beside the MFMAs here.  gnp_raw (the un-normalised operand copy a later 1x1 shortcut reads) is
// written by the first column tile from the same registers.
// ---------------------------------------------------------------------------
template <typename TM, int LTH> struct GnInloop {
  static constexpr int EPC = 16 / (int)sizeof(TM);                          // operand elements per 16-B piece
  static constexpr int BKE = 8 * EPC;                                       // channels per chunk (a 128-B tile row)
  static constexpr int QPR = BKE / 4;                                       // float4 quads per chunk row
  static constexpr int RPQ = LTH / QPR;                                     // panel rows per pass of the LTH producer threads
  static constexpr int NP = 128 / RPQ;                                      // passes = rows per thread and chunk
  static constexpr int NX = NP + 8;                                         // loads per thread and chunk: rows + gamma, beta + 3 x (scale, shift)
  static constexpr int TAB_BYTES = 4096;                                    // (mean, rstd) table [3][8] float2 | block sums [3][64] double2
  static_assert(LTH % QPR == 0 && 128 % RPQ == 0, "producer geometry");
  struct XSet { float4 x[NP], ga, be, ts[3], tf[3]; };

  int quad, prow0, b_lo, nbi, Cg, C, c0a;
  int roff[NP];                                                             // b * T + t of panel row prow0 + k * RPQ, or -1 (pad row / outside the tensor)
  unsigned items;                                                           // 2 bits per row: its batch item - b_lo

  // per-thread constants.  ltid = index among the LTH producer threads; panel row p holds padded row q0 - 1 + p
  __device__ __forceinline__ void setup(const GemmArgs& g, int q0, int ltid, int rlo, int rhi) {
    const int T = g.Tin, P = T + 1, MP = g.B * P;
    C = g.c0; c0a = C - g.gnp_c1; Cg = C / g.gnp_G;
    quad = ltid % QPR; prow0 = ltid / QPR;
    b_lo = rlo / T;
    nbi = (rhi - 1) / T - b_lo + 1;
    items = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int qi = q0 - 1 + prow0 + k * RPQ;
      const int b = qi > 0 ? qi / P : 0;
      const int t = qi - b * P;
      const bool ok = qi >= 0 && qi < MP && t < T;
      roff[k] = ok ? qi - b : -1;
      items |= (unsigned)(ok ? min(max(b - b_lo, 0), 2) : 0) << (2 * k);
    }
  }
  // (mean, rstd) of every (item, group) the tile touches -> tab[item * 8 + group].  Every thread of the workgroup calls it (two barriers).
  // Same finalisation as GnPrologue::finish / gn_apply_kernel: the block sums are exact, their order is free.
  __device__ __forceinline__ void table(const GemmArgs& g, int tid, char* tabmem) {
    const int T = g.Tin, G = g.gnp_G;
    float2* const gtab = reinterpret_cast<float2*>(tabmem);
    double2* const bsum = reinterpret_cast<double2*>(tabmem + 256);
    const int nblk = C >> 4, nblk0 = c0a >> 4;
    if (tid < nbi * nblk) {
      const int bi = tid / nblk, blk = tid - bi * nblk;
      const long long* st = blk < nblk0 ? g.gnp_stats + ((size_t)(b_lo + bi) * nblk0 + blk) * 2
                                        : g.gnp_stats1 + ((size_t)(b_lo + bi) * (nblk - nblk0) + (blk - nblk0)) * 2;
      bsum[tid] = make_double2((double)st[0] * (1.0 / GN_SUM_SCALE), (double)st[1] * (1.0 / GN_SQ_SCALE));
    }
    __syncthreads();
    if (tid < nbi * G) {
      const int bi = tid / G, gq = tid - bi * G;
      const int nb = Cg >> 4;
      double ds = 0.0, dq = 0.0;
      for (int j = 0; j < nb; ++j) { const double2 e = bsum[bi * nblk + gq * nb + j]; ds += e.x; dq += e.y; }
      const float inv_nf = 1.0f / ((float)T * (float)Cg);
      const double inv_n = (double)inv_nf * (2.0 - (double)inv_nf * ((double)T * (double)Cg));
      const double mean = ds * inv_n;
      double var = dq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float ve = (float)var + g.gnp_eps;
      float r = rsqrtf(ve);
      r = r * (1.5f - 0.5f * ve * r * r);
      gtab[bi * 8 + gq] = make_float2((float)mean, r);
    }
    __syncthreads();
  }
  // the NX loads of chunk ch (always NX of them, from clamped addresses: the counted waits rely on the number)
  __device__ __forceinline__ void load(const GemmArgs& g, int ch, XSet& s) const {
    const int cq = ch * BKE + quad * 4;
    const float* xs; int xld;
    if (cq < c0a) { xs = g.gnp_x + cq; xld = g.gnp_ldx; } else { xs = g.gnp_x1 + (cq - c0a); xld = g.gnp_ldx1; }
#pragma unroll
    for (int k = 0; k < NP; ++k) s.x[k] = *reinterpret_cast<const float4*>(xs + (size_t)max(roff[k], 0) * xld);
    s.ga = *reinterpret_cast<const float4*>(g.gnp_gamma + cq);
    s.be = *reinterpret_cast<const float4*>(g.gnp_beta + cq);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // (item clamped to the tile's last one; without a time embedding the loads go to gamma: same count, values unused)
      const float* tp = g.gnp_temb ? g.gnp_temb + (size_t)(b_lo + min(i, nbi - 1)) * g.gnp_ldtemb + cq : g.gnp_gamma + cq;
      const float* tq = g.gnp_temb ? tp + C : tp;
      if (((reinterpret_cast<uintptr_t>(tp) | reinterpret_cast<uintptr_t>(tq)) & 15) == 0) {
        s.ts[i] = *reinterpret_cast<const float4*>(tp);
        s.tf[i] = *reinterpret_cast<const float4*>(tq);
      } else {                                                              // (an odd channel count / row stride: still two "loads" for the count's sake are 8 here -- the
        s.ts[i] = make_float4(tp[0], tp[1], tp[2], tp[3]);                  //  counted waits then only wait longer than necessary)
        s.tf[i] = make_float4(tq[0], tq[1], tq[2], tq[3]);
      }
    }
  }
  // act(x * a + b) of the loaded chunk -> ring slot `slot` (LDS address of its 16 KB), rows in the swizzled image the consumers read:
  // byte r * 128 + ((c16 ^ ((r >> 1) & 7)) * 16) holds the 16-B piece c16 of panel row r.  `raw`: also store the un-normalised rows.
  __device__ __forceinline__ void produce(const GemmArgs& g, int ch, const XSet& s, char* slot, const char* tabmem, bool raw) const {
    const int cq = ch * BKE + quad * 4;
    const int gg = cq / Cg;
    float a[3][4], b[3][4];
    const float gam[4] = {s.ga.x, s.ga.y, s.ga.z, s.ga.w}, bet[4] = {s.be.x, s.be.y, s.be.z, s.be.w};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < nbi) {
        float2 mr = reinterpret_cast<const float2*>(tabmem)[i * 8 + gg];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // (the packed-product hazard of GnPrologue::affine: the pair lands before its first use)
        asm volatile("" : "+v"(mr.x), "+v"(mr.y));
        const float ts[4] = {s.ts[i].x, s.ts[i].y, s.ts[i].z, s.ts[i].w}, tf[4] = {s.tf[i].x, s.tf[i].y, s.tf[i].z, s.tf[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[i][e] = mr.y * gam[e];
          b[i][e] = bet[e] - mr.x * a[i][e];
          if (g.gnp_temb) {
            const float s1 = 1.0f + ts[e];
            a[i][e] *= s1;
            b[i][e] = b[i][e] * s1 + tf[e];
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[i][e] = a[0][e]; b[i][e] = b[0][e]; }
      }
    }
    TM* const rawp = reinterpret_cast<TM*>(g.gnp_raw);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int p = prow0 + k * RPQ;
      const int it = (items >> (2 * k)) & 3;
      const float w4[4] = {s.x[k].x, s.x[k].y, s.x[k].z, s.x[k].w};
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ae = it == 0 ? a[0][e] : it == 1 ? a[1][e] : a[2][e];
        const float be_ = it == 0 ? b[0][e] : it == 1 ? b[1][e] : b[2][e];
        y[e] = w4[e] * ae + be_;
      }
      if (g.gnp_silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = silu_f(y[e]);
      }
      const bool ok = roff[k] >= 0;
      if (!ok) { y[0] = y[1] = y[2] = y[3] = 0.f; }
      if constexpr (sizeof(TM) == 4) {
        const int phys = quad ^ ((p >> 1) & 7);
        *reinterpret_cast<float4*>(slot + p * 128 + phys * 16) = make_float4(y[0], y[1], y[2], y[3]);
      } else {
        const int phys = (quad >> 1) ^ ((p >> 1) & 7);
        uint2 v;
        if constexpr (std::is_same<TM, f16_t>::value) { v.x = pack_f16x2(y[0], y[1]); v.y = pack_f16x2(y[2], y[3]); }
        else { v.x = pack_bf16x2(y[0], y[1]); v.y = pack_bf16x2(y[2], y[3]); }
        *reinterpret_cast<uint2*>(slot + p * 128 + phys * 16 + (quad & 1) * 8) = v;
      }
      if (raw && ok) out_op4<TM>(rawp + (size_t)roff[k] * g.lda0 + cq, w4[0], w4[1], w4[2], w4[3]);
    }
  }
};

}  // namespace ns2vc
