// Implicit-GEMM kernel for every Conv1d (k=3 / k=1, stride 1 / stride 2 /
// nearest-upsample-then-conv) and Linear of the NS2VC denoiser, CDNA4 (gfx950).
//
// Replaces the reference's F.conv1d / F.linear call sites
// (unet1d/lora.py:98-104,119-123; unet1d/resnet.py:591-641, 138-173, 214-223;
//  unet1d/attention.py:206-301; unet1d/attention_processor.py:1013-1045).
//
//   out[m][n] = epi( sum_k A[m,k] * W[n][k] ),  m = b*Tout + t,  k = tap*(c0+c1) + c
//
// A is an "operand tensor": channels-last [B][Tin][C] already in the MFMA operand
// type (bf16, or fp32 in parity mode) and already normalised/activated by the
// producing kernel (or raw, with the LayerNorm applied by linearity in the epilogue),
// so BOTH operands stream HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B
// per lane, no VGPR round trip, no VALU in the main loop):
//   * conv taps, stride 2 and nearest-upsample are per-lane SOURCE row offsets computed
//     once per (piece, tap, source tensor); zero padding / rows past M are out-of-range
//     offsets of the buffer descriptor (the DMA writes zeros); the K position of a tile
//     is the instruction's scalar offset; the skip-connection concat is a K loop over
//     two descriptors (no torch.cat copy), a fused 1x1 shortcut a third;
//   * the LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is
//     applied to the source address: 16-B chunk c of tile row r is stored at chunk
//     position c ^ ((r>>1)&7) and read back with the same XOR -> conflict-free
//     ds_read_b128 for the 32x32 MFMA fragment reads;
//   * STAGES-deep LDS ring, counted s_waitcnt vmcnt(N) (never 0 in steady state),
//     one raw s_barrier per K tile; the DMA is issued through inline asm so the
//     compiler does not drain it in front of every ds_read.
// Kernels: gemm4_kernel (default; 512 threads, two wave groups on opposite halves of every
// K tile, see its header) and gemm2_kernel (256 threads = 2x2 waves; narrow GEGLU and the
// 64-column fallback).  Wave tiles are built from 32x32 MFMA tiles:
//   fp16: v_mfma_f32_32x32x16_f16  (one per 32-B k-slab; the default 16-bit mode: 8.3e-4 end to end)
//   bf16: v_mfma_f32_32x32x16_bf16 (one per 32-B k-slab; same rate and bytes, 6.5e-3 end to end)
//   f32 : v_mfma_f32_32x32x2_f32   (four per 32-B k-slab; exact fp32 parity mode)
// Epilogue: bias, GEGLU (value * gelu_erf(gate)), LayerNorm fix-up, fp32 residual add,
// fp32 store (residual stream) and/or operand-typed store (feeds the next GEMM /
// attention), GroupNorm / LayerNorm partial statistics of the result.
#include "common.h"
#include "mma.h"

namespace ns2vc {

// optional per-workgroup phase timestamps (s_memtime) for tuning: [block][8] uint64, set by ns2vc_debug_set_gemm_trace
// (compiled in only with -DNS2VC_GEMM_TRACE=1: `make TRACE=1`; the stamps cost a few % otherwise)
__device__ unsigned long long* g_gemm_trace = nullptr;
#ifndef NS2VC_GEMM_TRACE
#define NS2VC_GEMM_TRACE 0
#endif
#if NS2VC_GEMM_TRACE
#define NS2VC_STAMP(i) do { if (tr && tid == 0) tr[i] = __builtin_readcyclecounter(); } while (0)
#define NS2VC_TRACE_PTR() (g_gemm_trace ? g_gemm_trace + (size_t)blockIdx.x * 8 : nullptr)
#else
#define NS2VC_STAMP(i) do { (void)tr; } while (0)
#define NS2VC_TRACE_PTR() nullptr
#endif

constexpr int TROW = 128;   // bytes of K per tile row (64 bf16 / 32 f32)

// ---------------------------------------------------------------------------
// LayerNorm by linearity.  LayerNorm(x) W^T = rstd * (x W^T - mean * rowsum(W)), so a GEMM whose input is a LayerNorm
// reads the RAW x (the operand copy its producer writes anyway) and fixes the result up in the epilogue; the
// producer's epilogue leaves (sum, sum of squares) per row and 64-column slice as plain fp32 stores (one writer per
// slot: deterministic, nothing to zero).  No normalisation pass over HBM.
// ---------------------------------------------------------------------------
// consumer, part 1 (top of the kernel, so the cold-load latency hides under the K loop): this lane's row pairs, raw
struct LnRaw { float4 v[4]; };                      // up to 8 slices of 64 channels = ln_dim 512
__device__ __forceinline__ void ln_row_load(const GemmArgs& g, int m, bool valid, LnRaw& r) {
  const int n4 = g.ln_stats ? (g.ln_dim >> 7) : 0;  // float4 = two (sum, sumsq) pairs = 128 channels
  const float4* p = reinterpret_cast<const float4*>(g.ln_stats + (size_t)min(m, g.M - 1) * (g.ln_dim >> 6) * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = (valid && i < n4) ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
// consumer, part 2 (epilogue): mean / rstd of the row
__device__ __forceinline__ void ln_row_finish(const GemmArgs& g, const LnRaw& r, float& mean_f, float& rstd_f, int n0) {
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { s += r.v[i].x + r.v[i].z; q += r.v[i].y + r.v[i].w; }
  const float inv = 1.0f / (float)max(g.ln_dim, 1);
  const float mean = s * inv;
  double var = (double)q * (double)inv - (double)mean * (double)mean;     // the one cancellation-prone step
  if (var < 0.0) var = 0.0;
  mean_f = mean;
  rstd_f = 1.0f / sqrtf((float)var + g.ln_eps);
  // health of the linearity trick: the 16-bit modes round the raw row BEFORE centring, so the error on a row grows with
  // |mean| / std.  The first column workgroup of every row panel reports the largest ratio it sees (a plain read first:
  // the atomic is issued only by a wave that raises the maximum, i.e. a handful of times per forward).
  if (g.ln_health && n0 == 0) {
    float ratio = fabsf(mean) * rstd_f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ratio = fmaxf(ratio, __shfl_xor(ratio, o));
    if ((threadIdx.x & 63) == 0 && ratio > __uint_as_float(__hip_atomic_load(g.ln_health, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
      atomicMax(g.ln_health, __float_as_uint(ratio));
  }
}
// sum over the 16 lanes (one DPP row) that hold one 64-column slice of a result row; no LDS traffic, all 16 get the total
__device__ __forceinline__ float sum16_dpp(float x) {
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));   // row_ror:4
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));   // row_ror:8
  return x;
}
// producer (64-column wave tiles only): (sum, sumsq) of this row's 64-column slice -> rowstats[m][ncol/64]
// (plain store: every slot has exactly one writer)
__device__ __forceinline__ void ln_row_store(const GemmArgs& g, int m, int ncol, int cq, float ps, float pq) {
  ps = sum16_dpp(ps); pq = sum16_dpp(pq);
  if (cq == 0 && m < g.M) *reinterpret_cast<float2*>(g.rowstats + ((size_t)m * (g.N >> 6) + (ncol >> 6)) * 2) = make_float2(ps, pq);
}

// ---------------------------------------------------------------------------
// shared epilogue (bias, GEGLU, residual, fp32 / operand stores, GroupNorm statistics)
// ---------------------------------------------------------------------------
template <typename TM, int BM, int BN, bool LNC>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16_t (&acc)[BM / 64][BN / 64], char* smem, int m0, int n0, int tid,
                                              unsigned long long* tr, const LnRaw& lnraw) {
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  // ---- epilogue.  C layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), i.e. a lane
  // owns ONE column: direct stores would be 4-byte (fp32) / 2-byte (bf16) scalars.  Instead every wave transposes
  // its tile through its own slice of the (now idle) LDS ring and then moves whole rows: 16-B loads of bias /
  // residual, 16-B fp32 and 8-B bf16 stores, fully coalesced.
  constexpr int EP = WN + 4;                       // LDS pitch in floats (16-B aligned rows)
  NS2VC_STAMP(4);
  __syncthreads();                                 // every wave is done reading the last K tile
  float* et = reinterpret_cast<float*>(smem) + wave * (WM * EP);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) et[(i * 32 + 8 * (r >> 2) + 4 * hi + (r & 3)) * EP + j * 32 + l31] = acc[i][j][r];
  __syncthreads();
  NS2VC_STAMP(5);
  float* of = g.out_f32;
  TM* oo = reinterpret_cast<TM*>(g.out_op);
  const int mw0 = m0 + wm * WM;
  // LayerNorm-by-linearity consumer: lane l holds mean / rstd of row mw0 + l (l < WM; the pairs were loaded at the top
  // of the kernel so the latency hid under the K loop), fetched per row by shuffle
  constexpr bool lnc = LNC;           // compile-time: GEMMs that are not LayerNorm consumers carry none of this
  float lmean = 0.f, lrstd = 1.f;
  if constexpr (lnc) ln_row_finish(g, lnraw, lmean, lrstd, n0);
  if (g.geglu) {
    if constexpr (NT == 2) {
      constexpr int LPR = 8, RPI = 8, NIT = WM / RPI;          // 32 output columns per row = 8 lanes x 4
      const int rsub = lane >> 3, cq = lane & 7;
      const int pcol = n0 + wn * WN + cq * 4;                  // packed column of the value quad; gate quad = +32
      const int ocol = ((n0 + wn * WN) >> 1) + cq * 4;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv, wsv = bv, wsg = bv;
      if (g.bias) { bv = *reinterpret_cast<const float4*>(g.bias + pcol); bg = *reinterpret_cast<const float4*>(g.bias + pcol + 32); }
      if constexpr (lnc) { wsv = *reinterpret_cast<const float4*>(g.ln_wsum + pcol); wsg = *reinterpret_cast<const float4*>(g.ln_wsum + pcol + 32); }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = it * RPI + rsub, m = mw0 + row;
        float4 a = *reinterpret_cast<const float4*>(et + row * EP + cq * 4);
        float4 t = *reinterpret_cast<const float4*>(et + row * EP + 32 + cq * 4);
        if constexpr (lnc) {
          const float mu = __shfl(lmean, row), rs = __shfl(lrstd, row);
          a.x = rs * (a.x - mu * wsv.x); a.y = rs * (a.y - mu * wsv.y); a.z = rs * (a.z - mu * wsv.z); a.w = rs * (a.w - mu * wsv.w);
          t.x = rs * (t.x - mu * wsg.x); t.y = rs * (t.y - mu * wsg.y); t.z = rs * (t.z - mu * wsg.z); t.w = rs * (t.w - mu * wsg.w);
        }
        if (m < g.M) {
          float4 v;
          v.x = (a.x + bv.x) * gelu_erf_f(t.x + bg.x); v.y = (a.y + bv.y) * gelu_erf_f(t.y + bg.y);
          v.z = (a.z + bv.z) * gelu_erf_f(t.z + bg.z); v.w = (a.w + bv.w) * gelu_erf_f(t.w + bg.w);
          if (g.res) {
            const float4 rr = *reinterpret_cast<const float4*>(g.res + (size_t)m * g.ldres + ocol);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (of) out_f4(of + (size_t)m * g.ldo_f32 + ocol, v.x, v.y, v.z, v.w);
          if (oo) out_op4<TM>(oo + (size_t)m * g.ldo_op + ocol, v.x, v.y, v.z, v.w);
        }
      }
      (void)LPR;
    }
  } else {
    constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = WM / RPI;
    const int rsub = lane / LPR, cq = lane % LPR;
    const int ncol = n0 + wn * WN + cq * 4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), ws = bv;
    if (g.bias) bv = *reinterpret_cast<const float4*>(g.bias + ncol);
    if constexpr (lnc) ws = *reinterpret_cast<const float4*>(g.ln_wsum + ncol);
    // optional GroupNorm statistics of the result: this wave's rows belong to batch b0 or b0+1 (Tout >= WM)
    const int b0 = g.stats ? min(mw0, g.M - 1) / g.Tout : 0;     // (division only when the statistics are wanted)
    const int mB = g.stats ? (b0 + 1) * g.Tout : 0x7fffffff;    // first row of the next batch item
    float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
    constexpr int RB = NIT < 8 ? NIT : 8;                      // residual rows fetched per batch (before any store:
#pragma unroll                                                 //  res may alias out_f32 element-for-element)
    for (int it0 = 0; it0 < NIT; it0 += RB) {
      float4 rr[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int m = mw0 + (it0 + k) * RPI + rsub;
        rr[k] = (g.res && m < g.M) ? *reinterpret_cast<const float4*>(g.res + (size_t)m * g.ldres + ncol) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int row = (it0 + k) * RPI + rsub, m = mw0 + row;
        float4 a = *reinterpret_cast<const float4*>(et + row * EP + cq * 4);
        if constexpr (lnc) {
          const float mu = __shfl(lmean, row), rs = __shfl(lrstd, row);
          a.x = rs * (a.x - mu * ws.x); a.y = rs * (a.y - mu * ws.y); a.z = rs * (a.z - mu * ws.z); a.w = rs * (a.w - mu * ws.w);
        }
        float ps = 0.f, pq = 0.f;
        if (m < g.M) {
          float4 v;
          v.x = a.x + bv.x + rr[k].x; v.y = a.y + bv.y + rr[k].y; v.z = a.z + bv.z + rr[k].z; v.w = a.w + bv.w + rr[k].w;
          if (of) out_f4(of + (size_t)m * g.ldo_f32 + ncol, v.x, v.y, v.z, v.w);
          if (oo) out_op4<TM>(oo + (size_t)m * g.ldo_op + ncol, v.x, v.y, v.z, v.w);
          ps = (v.x + v.y) + (v.z + v.w); pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          if (m < mB) { gs0 += ps; gq0 += pq; } else { gs1 += ps; gq1 += pq; }
        }
        if constexpr (LPR == 16) { if (g.rowstats) ln_row_store(g, m, ncol, cq, ps, pq); }
      }
    }
    if (g.stats) {
      // fixed shuffle tree over the lanes that share a 16-channel block (4 column quads x all row lanes),
      // then ONE int64 fixed-point atomic per (batch item, block, moment): order-independent => deterministic
      double d0 = gs0, d1 = gq0, d2 = gs1, d3 = gq1;
#pragma unroll
      for (int o = 1; o <= 2; o <<= 1) {                       // the 4 column quads of a 16-channel block
        d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o);
      }
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {                     // the row lanes
        d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o);
      }
      if (rsub == 0 && (cq & 3) == 0 && mw0 < g.M) {
        const int blk = ncol >> 4, nblk = g.N >> 4;
        unsigned long long* st = reinterpret_cast<unsigned long long*>(g.stats) + ((size_t)b0 * nblk + blk) * 2;
        atomicAdd(st, (unsigned long long)llrint(d0 * GN_SUM_SCALE));
        atomicAdd(st + 1, (unsigned long long)llrint(d1 * GN_SQ_SCALE));
        if (mB < g.M && mB < mw0 + WM) {
          atomicAdd(st + 2 * nblk, (unsigned long long)llrint(d2 * GN_SUM_SCALE));
          atomicAdd(st + 2 * nblk + 1, (unsigned long long)llrint(d3 * GN_SQ_SCALE));
        }
      }
    }
  }
#if NS2VC_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (only so that the last stamp includes the store drain)
#endif
  NS2VC_STAMP(6);
}

template <typename TM, int BM, int BN, int STAGES, bool LNC>
__global__ __launch_bounds__(256) void gemm2_kernel(const GemmArgs g, const int flags) {
  op_mode_init<TM>();
  constexpr int EPC = MmaT<TM>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int LA = BM / 32, LB = BN / 32, LPT = LA + LB;     // 16-B DMA pieces per thread per tile
  constexpr int STAGE = (BM + BN) * TROW;
  static_assert(LPT * (STAGES - 1) < 60, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned long long* tr = NS2VC_TRACE_PTR();
  NS2VC_STAMP(0);

  // ---- XCD-aware tile mapping (blocks sharing an activation row-panel sit on one XCD's L2)
  const int nb_n = g.N / BN;
  const int nb_m = (g.M + BM - 1) / BM;
  const int nwg = nb_n * nb_m;
  int tm, tn;
  {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tm = swz / nb_n;
    tn = swz - tm * nb_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  LnRaw lnraw;                         // LayerNorm-by-linearity consumer: this lane's row pairs (see gemm_epilogue)
  if constexpr (LNC) ln_row_load(g, m0 + wm * (BM / 2) + lane, lane < BM / 2, lnraw);

  // ---- per-thread DMA coordinates: piece i = tid + 256*j covers tile row i>>3, physical chunk i&7.  Buffer-descriptor
  // DMA as in gemm4_kernel: per-lane byte offsets computed once per (piece, tap, source tensor), the K position of a
  // tile in an SGPR, padded taps / rows past M as out-of-range offsets (the DMA writes zeros).
  const int prow = tid >> 3;                 // row inside a 32-row pass
  const int pchunk = tid & 7;
  const int Ctot = g.c0 + g.c1;
  const int smul = g.tmode == TMODE_DOWN2 ? 2 : 1;
  const int toff = g.taps >> 1;
  const int ulim = g.tmode == TMODE_UP2 ? g.Tout : g.Tin;
  const int ushr = g.tmode == TMODE_UP2 ? 1 : 0;
  constexpr unsigned SZB = sizeof(TM);
  unsigned p0t0[LA], p0t1[LA], p0t2[LA], p1t0[LA], p1t1[LA], p1t2[LA], p2c[LA];    // [source tensor][tap] byte offsets
  const bool plain = g.taps == 1 && g.tmode == TMODE_SAME && g.c1 == 0 && g.c2 == 0;   // a linear: source row == output row
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int row = j * 32 + prow;
    const int m = m0 + row;
    const unsigned acolb = (unsigned)((pchunk ^ ((row >> 1) & 7)) * EPC) * SZB;     // source-side swizzle, bytes
    const bool mok = m < g.M;
    if (plain) {          // (wave-uniform) no integer division, one offset instead of seven
      p0t0[j] = mok ? (unsigned)m * (unsigned)g.lda0 * SZB + acolb : DMA_OOB;
      p0t1[j] = p0t2[j] = p1t0[j] = p1t1[j] = p1t2[j] = p2c[j] = DMA_OOB;
      continue;
    }
    const int b = mok ? m / g.Tout : 0;
    const int t = m - b * g.Tout;
    auto src_row = [&](int tp) __attribute__((always_inline)) {
      const int u = t * smul + tp - toff;
      const bool ok = mok && (tp < g.taps) && (u >= 0) && (u < ulim);
      return ok ? b * g.Tin + min(u >> ushr, g.Tin - 1) : -1;
    };
    auto off = [&](int r, int ld) __attribute__((always_inline)) { return r >= 0 ? (unsigned)r * (unsigned)ld * SZB + acolb : DMA_OOB; };
    const int r0 = src_row(0), r1 = src_row(1), r2 = src_row(2);
    p0t0[j] = off(r0, g.lda0); p0t1[j] = off(r1, g.lda0); p0t2[j] = off(r2, g.lda0);
    p1t0[j] = off(r0, g.lda1); p1t1[j] = off(r1, g.lda1); p1t2[j] = off(r2, g.lda1);
    p2c[j] = off(toff == 0 ? r0 : r1, g.lda2);
  }
  unsigned vw[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int row = j * 32 + prow;
    vw[j] = ((unsigned)(n0 + row) * (unsigned)g.K + (unsigned)((pchunk ^ ((row >> 1) & 7)) * EPC)) * SZB;
  }
  const unsigned long long rowsA = (unsigned long long)g.B * g.Tin;
  const i32x4_t rA0 = make_rsrc(g.a0, rowsA * g.lda0 * SZB);
  const i32x4_t rA1 = make_rsrc(g.c1 ? g.a1 : g.a0, rowsA * (g.c1 ? g.lda1 : g.lda0) * SZB);
  const i32x4_t rA2 = make_rsrc(g.c2 ? g.a2 : g.a0, rowsA * (g.c2 ? g.lda2 : g.lda0) * SZB);
  const i32x4_t rW = make_rsrc(g.w, (unsigned long long)g.N * g.K * SZB);

  const int K1 = g.taps * Ctot;                      // K of the main (conv / linear) segment; the rest is the fused 1x1 segment
  auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
    const int k0 = kt * BKE;
    const bool seg2 = k0 >= K1;                      // every branch here is wave-uniform
    const int k1 = seg2 ? 0 : k0;
    const int tap = k1 / Ctot;
    const int cc = k1 - tap * Ctot;
    const unsigned sbase = lds0 + stage * STAGE + wave * 1024;
    if (seg2) {
      const unsigned so = (unsigned)(k0 - K1) * SZB;
#pragma unroll
      for (int j = 0; j < LA; ++j) blds16(rA2, p2c[j], so, sbase + j * 4096);
    } else if (cc < g.c0) {
      const unsigned so = (unsigned)cc * SZB;
      if (tap == 0) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t0[j], so, sbase + j * 4096);
      } else if (tap == 1) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t1[j], so, sbase + j * 4096);
      } else {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t2[j], so, sbase + j * 4096);
      }
    } else {
      const unsigned so = (unsigned)(cc - g.c0) * SZB;
      if (tap == 0) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t0[j], so, sbase + j * 4096);
      } else if (tap == 1) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t1[j], so, sbase + j * 4096);
      } else {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t2[j], so, sbase + j * 4096);
      }
    }
    const unsigned bbase = sbase + BM * TROW;
    const unsigned soffW = (unsigned)k0 * SZB;
#pragma unroll
    for (int j = 0; j < LB; ++j) blds16(rW, vw[j], soffW, bbase + j * 4096);
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BKE;
  // experiment (flags bit 0): every row-panel starts its K loop at a different tile, so the workgroups of one
  // column do not all pull the same weight lines from L2 at the same moment
  const int rot = (flags & 1) ? tm % nk : 0;
  auto ktile = [&](int kt) __attribute__((always_inline)) { const int kk = kt + rot; return kk >= nk ? kk - nk : kk; };
  NS2VC_STAMP(1);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue_tile(ktile(s), s);
  NS2VC_STAMP(2);
  const int l31 = lane & 31, hi = lane >> 5;
  // fragment reads: row r = w*W? + i*32 + l31, logical chunk 2*ks+hi stored at chunk ^ ((r>>1)&7);
  // (r>>1)&7 == (l31>>1)&7 because every fragment row block starts at a multiple of 32
  const int sw = (l31 >> 1) & 7;
  int stage = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed when at most (tiles issued after it) * LPT pieces are still in flight
    const int after = min(STAGES - 2, nk - 1 - kt);
    if (STAGES >= 4 && after >= 2) wait_vmcnt<2 * LPT>();
    else if (after >= 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my ds_reads of the stage about to be refilled are done
    __builtin_amdgcn_s_barrier();
    if (kt == 0) NS2VC_STAMP(3);
    if (kt + STAGES - 1 < nk && !(flags & 4)) {      // (flags bit 2: ablation, no steady-state loads)
      int st2 = stage + STAGES - 1;
      if (st2 >= STAGES) st2 -= STAGES;
      issue_tile(ktile(kt + STAGES - 1), st2);
    }
    if (flags & 2) { if (++stage == STAGES) stage = 0; continue; }   // (flags bit 1: ablation, loads only)
    const char* As = smem + stage * STAGE;
    const char* Bs = As + BM * TROW;
    const char* ap = As + (wm * WM + l31) * TROW;
    const char* bp = Bs + (wn * WN + l31) * TROW;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + hi) ^ sw) * 16;
      u32x4_t af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(ap + i * 32 * TROW + coff);
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(bp + j * 32 * TROW + coff);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) MmaT<TM>::mma(acc[i][j], af[i], bf[j]);
    }
    if (++stage == STAGES) stage = 0;
  }
  gemm_epilogue<TM, BM, BN, LNC>(g, acc, smem, m0, n0, tid, tr, lnraw);
}

// ---------------------------------------------------------------------------
// epilogue of the 8-wave K-split kernel (gemm4_kernel): per 32-row slab, both K halves stage their
// partial tile in LDS (re-using the operand ring), then each of the eight waves adds the pair for 16 rows and moves
// whole rows out (16-B fp32 / 8-B bf16 stores, coalesced); bias, GEGLU, LayerNorm fix-up, residual, statistics.
// ---------------------------------------------------------------------------
template <typename TM, int BM, bool LNC>
__device__ __forceinline__ void gemm4_epilogue(const GemmArgs& g, f32x16_t (&acc)[BM / 64][2], char* smem, int m0, int n0, int tid,
                                               unsigned long long* tr, const LnRaw& lnraw) {
  constexpr int WM = BM / 2, WN = 64, MT = WM / 32, NT = 2;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, wq = wave & 3;
  const int wm = wq >> 1, wn = wq & 1;
  const int l31 = lane & 31, hi = lane >> 5;
  (void)NT;
  NS2VC_STAMP(4);
  // ---- epilogue: per 32-row slab, both K halves stage their partial tile, then each of the eight waves adds the
  // pair for 16 rows and moves whole rows out (16-B fp32 / 8-B bf16 stores, coalesced)
  constexpr int EP = WN + 4;                        // staging pitch in floats
  constexpr int SLAB = 32 * EP;                     // floats per staged 32 x 64 slab
  float* const et_mine = reinterpret_cast<float*>(smem) + wave * SLAB;
  const float* const et_a = reinterpret_cast<const float*>(smem) + wq * SLAB + kg * 16 * EP;        // K half 0, my 16 rows
  const float* const et_b = et_a + 4 * SLAB;                                                          // K half 1
  float* of = g.out_f32;
  TM* oo = reinterpret_cast<TM*>(g.out_op);
  const int mw0 = m0 + wm * WM;                     // first row of the wave tile (both K halves)
  const int b0 = g.stats ? min(mw0, g.M - 1) / g.Tout : 0;       // (division only when the statistics are wanted)
  const int mB = g.stats ? (b0 + 1) * g.Tout : 0x7fffffff;
  float gs0 = 0.f, gq0 = 0.f, gs1 = 0.f, gq1 = 0.f;
  constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = 16 / RPI;    // 16 lanes per row, 4 rows per pass, 4 passes
  const int rsub = lane / LPR, cq = lane % LPR;
  const int ncol = n0 + wn * WN + cq * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias && !g.geglu) bv = *reinterpret_cast<const float4*>(g.bias + ncol);
  // GEGLU: 32 output columns per row = 8 lanes x 4, 8 rows per pass
  const int grsub = lane >> 3, gcq = lane & 7;
  const int pcol = n0 + wn * WN + gcq * 4;          // packed column of the value quad; gate quad = +32
  const int ocol = ((n0 + wn * WN) >> 1) + gcq * 4;
  float4 gbv = make_float4(0.f, 0.f, 0.f, 0.f), gbg = gbv;
  if (g.bias && g.geglu) { gbv = *reinterpret_cast<const float4*>(g.bias + pcol); gbg = *reinterpret_cast<const float4*>(g.bias + pcol + 32); }
  // LayerNorm-by-linearity consumer: lane l < 16*MT holds mean / rstd of the l-th row this wave will emit (loaded before the K loop)
  constexpr bool lnc = LNC;
  float lmean = 0.f, lrstd = 1.f;
  float4 ws = make_float4(0.f, 0.f, 0.f, 0.f), wsv = ws, wsg = ws;
  if constexpr (lnc) {
    ln_row_finish(g, lnraw, lmean, lrstd, n0);
    if (g.geglu) { wsv = *reinterpret_cast<const float4*>(g.ln_wsum + pcol); wsg = *reinterpret_cast<const float4*>(g.ln_wsum + pcol + 32); }
    else ws = *reinterpret_cast<const float4*>(g.ln_wsum + ncol);
  }

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    lds_barrier();                                  // ring (or the previous slab) is free.  (LDS-only barriers: the stores of
                                                    //  the previous slab are in flight and nobody here waits for them)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) et_mine[(8 * (r >> 2) + 4 * hi + (r & 3)) * EP + j * 32 + l31] = acc[mt][j][r];
    lds_barrier();
    if (mt == 0) NS2VC_STAMP(5);
    const int mrow0 = mw0 + mt * 32 + kg * 16;      // first of my 16 rows
    if (g.geglu) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + grsub, m = mrow0 + row;
        const float4 a0 = *reinterpret_cast<const float4*>(et_a + row * EP + gcq * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(et_b + row * EP + gcq * 4);
        const float4 t0 = *reinterpret_cast<const float4*>(et_a + row * EP + 32 + gcq * 4);
        const float4 t1 = *reinterpret_cast<const float4*>(et_b + row * EP + 32 + gcq * 4);
        float4 a = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        float4 t = make_float4(t0.x + t1.x, t0.y + t1.y, t0.z + t1.z, t0.w + t1.w);
        if constexpr (lnc) {
          const float mu = __shfl(lmean, mt * 16 + row), rs = __shfl(lrstd, mt * 16 + row);
          a.x = rs * (a.x - mu * wsv.x); a.y = rs * (a.y - mu * wsv.y); a.z = rs * (a.z - mu * wsv.z); a.w = rs * (a.w - mu * wsv.w);
          t.x = rs * (t.x - mu * wsg.x); t.y = rs * (t.y - mu * wsg.y); t.z = rs * (t.z - mu * wsg.z); t.w = rs * (t.w - mu * wsg.w);
        }
        if (m < g.M) {
          float4 v;
          v.x = (a.x + gbv.x) * gelu_erf_f(t.x + gbg.x); v.y = (a.y + gbv.y) * gelu_erf_f(t.y + gbg.y);
          v.z = (a.z + gbv.z) * gelu_erf_f(t.z + gbg.z); v.w = (a.w + gbv.w) * gelu_erf_f(t.w + gbg.w);
          if (g.res) {
            const float4 rr = *reinterpret_cast<const float4*>(g.res + (size_t)m * g.ldres + ocol);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (of) out_f4(of + (size_t)m * g.ldo_f32 + ocol, v.x, v.y, v.z, v.w);
          if (oo) out_op4<TM>(oo + (size_t)m * g.ldo_op + ocol, v.x, v.y, v.z, v.w);
        }
      }
    } else {
      float4 rr[NIT];                               // residual rows first (res may alias out_f32 element-for-element)
      if (g.res) {                                  // (uniform branch, rows past M read row M-1 and are never stored: straight-line
#pragma unroll                                      //  loads -- per-lane conditional loads were compiled with a wait after each)
        for (int k = 0; k < NIT; ++k) {
          const int m = min(mrow0 + k * RPI + rsub, g.M - 1);
          rr[k] = *reinterpret_cast<const float4*>(g.res + (size_t)m * g.ldres + ncol);
        }
      } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k) rr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // Two passes: FIRST every value of the slab is computed into its own registers (this consumes all residual rows and
      // LDS reads), THEN all stores are issued back to back.  Interleaved, the compiler had to wait for stores to complete
      // (s_waitcnt vmcnt) before it could reuse a store's data registers for the next row pass, and with loads and stores
      // both pending it can only wait with vmcnt(0): every pass sat through the write latency of the previous one.
      float4 vv[NIT];
      float2 rs2[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int row = k * RPI + rsub, m = mrow0 + row;
        const float4 a0 = *reinterpret_cast<const float4*>(et_a + row * EP + cq * 4);
        const float4 a1 = *reinterpret_cast<const float4*>(et_b + row * EP + cq * 4);
        float4 a = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
        if constexpr (lnc) {
          const float mu = __shfl(lmean, mt * 16 + row), rs = __shfl(lrstd, mt * 16 + row);
          a.x = rs * (a.x - mu * ws.x); a.y = rs * (a.y - mu * ws.y); a.z = rs * (a.z - mu * ws.z); a.w = rs * (a.w - mu * ws.w);
        }
        float ps = 0.f, pq = 0.f;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < g.M) {
          v.x = a.x + bv.x + rr[k].x; v.y = a.y + bv.y + rr[k].y; v.z = a.z + bv.z + rr[k].z; v.w = a.w + bv.w + rr[k].w;
          ps = (v.x + v.y) + (v.z + v.w); pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          if (m < mB) { gs0 += ps; gq0 += pq; } else { gs1 += ps; gq1 += pq; }
        }
        vv[k] = v;
        rs2[k] = make_float2(0.f, 0.f);
        if (g.rowstats) rs2[k] = make_float2(sum16_dpp(ps), sum16_dpp(pq));
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int m = mrow0 + k * RPI + rsub;
        if (m < g.M) {
          if (of) out_f4(of + (size_t)m * g.ldo_f32 + ncol, vv[k].x, vv[k].y, vv[k].z, vv[k].w);
          if (oo) out_op4<TM>(oo + (size_t)m * g.ldo_op + ncol, vv[k].x, vv[k].y, vv[k].z, vv[k].w);
          if (g.rowstats && cq == 0) *reinterpret_cast<float2*>(g.rowstats + ((size_t)m * (g.N >> 6) + (ncol >> 6)) * 2) = rs2[k];
        }
      }
    }
  }
  if (g.stats) {
    double d0 = gs0, d1 = gq0, d2 = gs1, d3 = gq1;
#pragma unroll
    for (int o = 1; o <= 2; o <<= 1) {
      d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o);
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      d0 += __shfl_xor(d0, o); d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); d3 += __shfl_xor(d3, o);
    }
    if (rsub == 0 && (cq & 3) == 0 && mw0 < g.M) {
      const int blk = ncol >> 4, nblk = g.N >> 4;
      unsigned long long* st = reinterpret_cast<unsigned long long*>(g.stats) + ((size_t)b0 * nblk + blk) * 2;
      atomicAdd(st, (unsigned long long)llrint(d0 * GN_SUM_SCALE));
      atomicAdd(st + 1, (unsigned long long)llrint(d1 * GN_SQ_SCALE));
      if (mB < g.M && mB < mw0 + WM) {
        atomicAdd(st + 2 * nblk, (unsigned long long)llrint(d2 * GN_SUM_SCALE));
        atomicAdd(st + 2 * nblk + 1, (unsigned long long)llrint(d3 * GN_SQ_SCALE));
      }
    }
  }
#if NS2VC_GEMM_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (only so that the last stamp includes the store drain)
#endif
  NS2VC_STAMP(6);
}

// ---------------------------------------------------------------------------
// 8-wave variant with an intra-workgroup K split.
//
// tools/dma_probe.hip (profiles/dma_probe_r01_*.txt) measured what feeds a CU on MI355X: one wave lands one 1-KB
// LDS-DMA piece per ~117 cycles however many it has in flight (4 waves: 35 B/clk/CU, 8 waves: 55-63 B/clk/CU = the L2
// peak), a 2-deep ring stalls for a full L2 round trip per K tile (3-deep streams continuously), and anything that
// misses the XCD's L2 arrives at <= 15 B/clk/CU.  The ablation runs of gemm2_kernel (tools/gemm_sweep.py --ablate) showed
// its K loop to be load-stream-bound at 17-23 B/clk/CU with the LDS fragment reads (1.5 KB per MFMA for a 32x64 wave
// tile) as the second limit.  This kernel is shaped by those numbers:
//   * 512 threads: waves 0-3 and 4-7 own the SAME 2x2 grid of 64x64 (or 32x64) wave tiles but opposite halves of every
//     64-wide K tile (ks 0,1 / ks 2,3), so a wave issues half the DMA pieces (3-4 per tile instead of 6-8), eight
//     waves keep the CU's load path full, and fragment reads drop to 0.5-0.75 KB per MFMA;
//   * BM x 128 tiles with BM = 128 where the grid still covers the chip (one workgroup per CU) and 64 otherwise;
//   * 3-deep ring: the DMA queue never drains between K tiles;
//   * the two K halves meet in the epilogue: both stage their accumulators in LDS (32-row slabs, re-using the ring),
//     then all eight waves add the pair while they transpose rows out -- every wave stores, nothing idles.
// ---------------------------------------------------------------------------
// Ablation build (make DEFS=-DNS2VC_GEMM_ABLATE=1 into a variant library, tools/gemm_sweep.py --ablate4): the K loop with one of
// its parts removed, to see what bounds it.  flags: 2 = loads only (no fragment reads, no MFMAs), 4 = no steady-state DMA
// (reads + MFMAs on whatever the ring holds), 8 = no fragment reads (MFMAs on registers read once), 16 = no MFMAs (DMA + reads)
#ifndef NS2VC_GEMM_ABLATE
#define NS2VC_GEMM_ABLATE 0
#endif
#ifndef NS2VC_G4_CONS_PF
#define NS2VC_G4_CONS_PF 0       // loader / consumer tiles: every fragment read of a K tile before its first MFMA (0: the compiler's order).
#endif                           // r5 session 2, same box: 4.054 vs 4.046 ms/step -- no gain (the tile is LDS-bandwidth-bound, not latency-bound), +19 VGPRs: off
#if NS2VC_GEMM_ABLATE
#define NS2VC_G4_FLAGS_PARAM , const int flags
#define NS2VC_G4_FLAG(b) ((flags & (b)) != 0)
#else
#define NS2VC_G4_FLAGS_PARAM
#define NS2VC_G4_FLAG(b) false
#endif
}  // namespace ns2vc
#include "gnpro.h"   // GnPrologue: act(GroupNorm(x)) of the rows a tile reads, built in front of its K loop
namespace ns2vc {

// SPEC (r3): loader / consumer wave specialisation.  profiles/r03_gemm_ablate4.txt: the DMA stream alone and the reads + MFMAs
// alone each take about half of the full loop's time -- they do not overlap, because every wave issues its DMA pieces (an
// LDS-DMA instruction holds the issuing wave until the CU's load path has taken its 1 KB: ~117 cycles per piece and wave) and
// only then its MFMAs, all eight waves in the same phase behind the per-tile barrier.  A second workgroup on the CU interleaves
// the phases by itself; the coarse levels (M = 3776 / 7520 rows: 118-354 tiles for 256 CUs) have none.  SPEC splits the roles:
//   1: waves 0-3 issue ALL DMA pieces and never multiply, waves 4-7 multiply the whole K tile and never load (512 threads)
//   2: 8 loader waves (the count the L2 -> LDS path needs for its 64 B/clk) + 8 consumer waves with the K split (1024 threads)
//   3: 8 loader waves + 4 consumer waves (768 threads)
// Loaders that have no role in the 8-wave epilogue leave after the K loop (s_barrier counts live waves only); the others bring
// zero accumulators, so the epilogue is unchanged.
template <int SPEC> struct G4Waves {
  static constexpr int NL = SPEC == 0 ? 8 : SPEC == 1 ? 4 : 8;       // waves that issue DMA
  static constexpr int NC = SPEC == 0 ? 8 : SPEC == 2 ? 8 : 4;       // waves that multiply
  static constexpr int NW = SPEC == 0 ? 8 : NL + NC;
};
// GNP: the instantiation that carries the GroupNorm prologue (its own kernels: the prologue's registers -- +20 at its peak -- would
// otherwise cost the 64-row loader / consumer tiles of EVERY GEMM their second workgroup per CU)
template <typename TM, int BM, int BN, int STAGES, bool LNC, int SPEC = 0, bool GNP = false>
__global__ __launch_bounds__(64 * G4Waves<SPEC>::NW) void gemm4_kernel(const GemmArgs g NS2VC_G4_FLAGS_PARAM) {
  op_mode_init<TM>();
  constexpr int EPC = MmaT<TM>::EPC;
  constexpr int BKE = 8 * EPC;
  constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NT = WN / 32;
  constexpr int NL = G4Waves<SPEC>::NL, NC = G4Waves<SPEC>::NC, NW = G4Waves<SPEC>::NW;
  constexpr int EOFF = NW - 8;                                  // first wave with a role in the 8-wave epilogue
  constexpr int LTH = NL * 64;                                  // threads that issue DMA
  constexpr int RPP = LTH / 8;                                  // tile rows per DMA pass (8 threads = one 128-B row)
  constexpr int PASSB = RPP * TROW;
  constexpr int LA = BM / RPP, LB = BN / RPP, LPT = LA + LB;    // 16-B DMA pieces per loading thread per tile
  constexpr int STAGE = (BM + BN) * TROW;
  static_assert(BN == 128 && NT == 2, "wave tile is (BM/2) x 64");
  static_assert(LPT * (STAGES - 1) < 60, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ewave = wave - EOFF;                   // role in the epilogue (< 0: none)
  const int kg = (ewave >> 2) & 1, wq = ewave & 3; // K half, wave tile
  const int wm = wq >> 1, wn = wq & 1;
  const bool loader = SPEC == 0 || wave < NL;
  const bool consumer = SPEC == 0 || wave >= NW - NC;
  const unsigned lds0 = (unsigned)(size_t)smem;
  unsigned long long* tr = NS2VC_TRACE_PTR();
  NS2VC_STAMP(0);

  const int nb_n = g.N / BN;
  const int nb_m = (g.M + BM - 1) / BM;
  const int nwg = nb_n * nb_m;
  int tm, tn;
  const bool coop = GNP && g.gnp_x != nullptr && g.gnp_sync != nullptr && nb_n > 1;     // (uniform over the grid; the launcher pads the grid for it)
  if (coop) {
    // cooperative GroupNorm prologue: whole row blocks per XCD, so that the column tiles that share a row block's rows also share an L2
    const int bid = blockIdx.x;
    const int q = nb_m >> 3, r = nb_m & 7, xcd = bid & 7, idx = bid >> 3;
    const int tml = idx / nb_n;
    if (tml >= q + (xcd < r ? 1 : 0)) return;                                      // padding of the last row block slot
    tm = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + tml;
    tn = idx - tml * nb_n;
  } else {
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tm = swz / nb_n;
    tn = swz - tm * nb_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  LnRaw lnraw;                         // LayerNorm-by-linearity consumer: pairs of the lane's epilogue row, in flight during the K loop
  if constexpr (LNC) if (ewave >= 0) ln_row_load(g, m0 + wm * (BM / 2) + (lane >> 4) * 32 + kg * 16 + (lane & 15), lane < 16 * (BM / 64), lnraw);

  // ---- DMA coordinates: piece j of this thread = tile row j*64 + tid/8, physical 16-B chunk tid%8.
  // Everything per-lane is a 32-bit byte offset computed ONCE (per piece, tap and source tensor); the K position of a
  // tile is a scalar added by the hardware (buffer addressing), rows in the zero padding / past M are out-of-range
  // offsets that the DMA turns into zeros.  The K loop carries no address arithmetic.
  const int prow = (tid & (LTH - 1)) >> 3, pchunk = tid & 7;
  constexpr unsigned SZB = sizeof(TM);
  const unsigned acolb = (unsigned)((pchunk ^ ((prow >> 1) & 7)) * EPC) * SZB;      // source-side swizzle, bytes
  unsigned vw[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) vw[j] = ((unsigned)(n0 + j * RPP + prow) * (unsigned)g.K) * SZB + acolb;
  const i32x4_t rW = make_rsrc(g.w, (unsigned long long)g.N * g.K * SZB);
  const int nk = g.K / BKE;
#ifndef NS2VC_G4_EARLY_B
#define NS2VC_G4_EARLY_B 1
#endif
  // The weight halves of the first tiles depend on nothing but the tile's column: they are in flight (cold: this layer's weights were
  // last touched a step ago) while the activation rows' offsets -- an integer division per piece -- are still being computed.
  if (NS2VC_G4_EARLY_B && loader) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) {
#pragma unroll
        for (int j = 0; j < LB; ++j) blds16(rW, vw[j], (unsigned)(s * BKE) * SZB, lds0 + s * STAGE + wave * 1024 + BM * TROW + j * PASSB);
      }
  }
  const bool gnp = GNP && g.gnp_x != nullptr;      // (uniform over the grid)
#ifndef NS2VC_GNP_XB128
#define NS2VC_GNP_XB128 NS2VC_GNP_XB
#endif
  GnPrologue<TM, (BM == 128 ? NS2VC_GNP_XB128 : NS2VC_GNP_XB)> gpro;     // (128-row tiles: one workgroup per CU by their LDS, so registers are free)
  // (measured r4, same box: issuing the prologue's loads up here, before the row-offset set-up, LOSES 1 % -- 3.962 vs 3.921 ms/step -- and
  //  plain instead of write-through stores of the rows change nothing, profiles/r04_ab_gn_prologue_variants.txt; both stay compile-time options)
#ifndef NS2VC_GNP_SPLIT
#define NS2VC_GNP_SPLIT 0
#endif
  if (NS2VC_GNP_SPLIT && gnp) gpro.begin(g, max(m0 - (g.taps >> 1), 0), min(m0 + BM + (g.taps >> 1), g.M), tm, tid, 64 * NW, coop ? tn : 0, coop ? nb_n : 1);
  const int Ctot = g.c0 + g.c1;
  const int smul = g.tmode == TMODE_DOWN2 ? 2 : 1;
  const int toff = g.taps >> 1;
  const int ulim = g.tmode == TMODE_UP2 ? g.Tout : g.Tin;
  const int ushr = g.tmode == TMODE_UP2 ? 1 : 0;
  unsigned p0t0[LA], p0t1[LA], p0t2[LA], p1t0[LA], p1t1[LA], p1t2[LA], p2c[LA];    // [source tensor][tap] byte offsets
  const bool plain = g.taps == 1 && g.tmode == TMODE_SAME && g.c1 == 0 && g.c2 == 0;   // a linear: source row == output row
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int m = m0 + j * RPP + prow;
    const bool mok = m < g.M;
    if (plain) {          // (wave-uniform) skips the integer division and six of the seven offsets: most launches are linears
      p0t0[j] = mok ? (unsigned)m * (unsigned)g.lda0 * SZB + acolb : DMA_OOB;
      p0t1[j] = p0t2[j] = p1t0[j] = p1t1[j] = p1t2[j] = p2c[j] = DMA_OOB;
      continue;
    }
    const int b = mok ? m / g.Tout : 0;
    const int t = m - b * g.Tout;
    auto src_row = [&](int tp) __attribute__((always_inline)) {       // source row of tap tp, or -1 (zero padding / past M)
      const int u = t * smul + tp - toff;
      const bool ok = mok && (tp < g.taps) && (u >= 0) && (u < ulim);
      return ok ? b * g.Tin + min(u >> ushr, g.Tin - 1) : -1;
    };
    auto off = [&](int row, int ld) __attribute__((always_inline)) { return row >= 0 ? (unsigned)row * (unsigned)ld * SZB + acolb : DMA_OOB; };
    const int r0 = src_row(0), r1 = src_row(1), r2 = src_row(2);
    p0t0[j] = off(r0, g.lda0); p0t1[j] = off(r1, g.lda0); p0t2[j] = off(r2, g.lda0);
    p1t0[j] = off(r0, g.lda1); p1t1[j] = off(r1, g.lda1); p1t2[j] = off(r2, g.lda1);
    p2c[j] = off(toff == 0 ? r0 : r1, g.lda2);                         // the fused 1x1 segment reads the centre tap's rows
  }
  const unsigned long long rowsA = (unsigned long long)g.B * g.Tin;
  const i32x4_t rA0 = make_rsrc(g.a0, rowsA * g.lda0 * SZB);
  const i32x4_t rA1 = make_rsrc(g.c1 ? g.a1 : g.a0, rowsA * (g.c1 ? g.lda1 : g.lda0) * SZB);
  const i32x4_t rA2 = make_rsrc(g.c2 ? g.a2 : g.a0, rowsA * (g.c2 ? g.lda2 : g.lda0) * SZB);

  // K-order walk state (tiles are issued strictly in order): tap, channel offset inside the tap, position in K
  int is_tap = 0, is_cc = 0, is_k = 0;
  const int K1 = g.taps * Ctot;
  auto issue_b = [&](int stage, int kt) __attribute__((always_inline)) {     // the weight half of tile kt alone
    if (!loader) return;
    const unsigned bbase = lds0 + stage * STAGE + wave * 1024 + BM * TROW;
#pragma unroll
    for (int j = 0; j < LB; ++j) blds16(rW, vw[j], (unsigned)(kt * BKE) * SZB, bbase + j * PASSB);
  };
  auto issue_tile = [&](int stage, bool with_b = true) __attribute__((always_inline)) {
    if (!loader) return;
    const unsigned sbase = lds0 + stage * STAGE + wave * 1024;
    // every branch below is wave-uniform: one fixed descriptor and one fixed offset register per DMA instruction
    if (is_k >= K1) {
      const unsigned so = (unsigned)(is_k - K1) * SZB;
#pragma unroll
      for (int j = 0; j < LA; ++j) blds16(rA2, p2c[j], so, sbase + j * PASSB);
    } else if (is_cc < g.c0) {
      const unsigned so = (unsigned)is_cc * SZB;
      if (is_tap == 0) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t0[j], so, sbase + j * PASSB);
      } else if (is_tap == 1) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t1[j], so, sbase + j * PASSB);
      } else {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA0, p0t2[j], so, sbase + j * PASSB);
      }
    } else {
      const unsigned so = (unsigned)(is_cc - g.c0) * SZB;
      if (is_tap == 0) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t0[j], so, sbase + j * PASSB);
      } else if (is_tap == 1) {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t1[j], so, sbase + j * PASSB);
      } else {
#pragma unroll
        for (int j = 0; j < LA; ++j) blds16(rA1, p1t2[j], so, sbase + j * PASSB);
      }
    }
    const unsigned bbase = sbase + BM * TROW;
    const unsigned soffW = (unsigned)is_k * SZB;
    if (with_b) {
#pragma unroll
      for (int j = 0; j < LB; ++j) blds16(rW, vw[j], soffW, bbase + j * PASSB);
    }
    is_k += BKE;
    is_cc += BKE;
    if (is_cc >= Ctot) { is_cc = 0; ++is_tap; }
  };

  f32x16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  NS2VC_STAMP(1);
  const bool early_b = NS2VC_G4_EARLY_B || gnp;
  if (gnp) {
    // the weight tiles do not depend on the prologue: in flight first, then the rows this tile reads are built (the table of
    // the prologue lives in the ring stage nobody has been issued into yet), then their DMA
    if (!NS2VC_G4_EARLY_B) {
#pragma unroll
      for (int s = 0; s < STAGES - 1; ++s)
        if (s < nk) issue_b(s, s);
    }
    if (!NS2VC_GNP_SPLIT) gpro.begin(g, max(m0 - toff, 0), min(m0 + BM + toff, g.M), tm, tid, 64 * NW, coop ? tn : 0, coop ? nb_n : 1);
    gpro.finish(g, tid, smem + (STAGES - 1) * STAGE);
  }
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue_tile(s, !early_b);
  NS2VC_STAMP(2);

  const int l31 = lane & 31, hi = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  int stage = 0;
#if NS2VC_GEMM_ABLATE
  u32x4_t abl_a[4][MT], abl_b[4][NT];
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int after = min(STAGES - 2, nk - 1 - kt);
    if (kt == 0 && early_b) {        // issue order was B(0) B(1) .. A(0) A(1) ..: tile 0 is complete when only the later A halves are out
      if (STAGES == 3 && after >= 1) wait_vmcnt<LA>(); else wait_vmcnt<0>();
    } else if (STAGES >= 4 && after >= 2) wait_vmcnt<2 * LPT>();
    else if (STAGES >= 3 && after >= 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt == 0) NS2VC_STAMP(3);
    auto refill = [&]() __attribute__((always_inline)) {
      if (kt + STAGES - 1 < nk && !NS2VC_G4_FLAG(4)) {
        int st2 = stage + STAGES - 1;
        if (st2 >= STAGES) st2 -= STAGES;
        issue_tile(st2);
      }
    };
    auto multiply = [&]() __attribute__((always_inline)) {
      if (NS2VC_G4_FLAG(2)) return;
      if (!consumer) return;
      const char* As = smem + stage * STAGE;
      const char* Bs = As + BM * TROW;
      const char* ap = As + (wm * WM + l31) * TROW;
      const char* bp = Bs + (wn * WN + l31) * TROW;
#if NS2VC_G4_CONS_PF && !NS2VC_GEMM_ABLATE
      if constexpr (NC == 4) {
        // r5: the loader / consumer tiles run ONE multiplying wave per SIMD, so nothing hides an LDS round trip: left alone the compiler
        // issues three fragment reads, waits, multiplies twice (found in convts.hip's ISA, same shape here).  Every fragment read of the K
        // tile first; the compiler's counted waits then release the MFMAs one k-slab at a time.  (With two multiplying waves per SIMD --
        // the plain K-split tiles -- the same change was measured as a loss in r2.)
        u32x4_t afa[4][MT], bfa[4][NT];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int coff = ((2 * kk + hi) ^ sw) * 16;
#pragma unroll
          for (int i = 0; i < MT; ++i) afa[kk][i] = *reinterpret_cast<const u32x4_t*>(ap + i * 32 * TROW + coff);
#pragma unroll
          for (int j = 0; j < NT; ++j) bfa[kk][j] = *reinterpret_cast<const u32x4_t*>(bp + j * 32 * TROW + coff);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) MmaT<TM>::mma(acc[i][j], afa[kk][i], bfa[kk][j]);
        return;
      }
#endif
#pragma unroll
      for (int kk = 0; kk < (NC == 8 ? 2 : 4); ++kk) {
        const int coff = ((2 * ((NC == 8 ? 2 * kg : 0) + kk) + hi) ^ sw) * 16;     // this K half's two 32-B k-slabs (4 consumer waves: all four)
        u32x4_t af[MT], bf[NT];
#if NS2VC_GEMM_ABLATE
        if (NS2VC_G4_FLAG(8) && kt > 0) {
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = abl_a[kk][i];
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[j] = abl_b[kk][j];
        } else
#endif
        {
#pragma unroll
          for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(ap + i * 32 * TROW + coff);
#pragma unroll
          for (int j = 0; j < NT; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(bp + j * 32 * TROW + coff);
        }
#if NS2VC_GEMM_ABLATE
        if (kt == 0) {
#pragma unroll
          for (int i = 0; i < MT; ++i) abl_a[kk][i] = af[i];
#pragma unroll
          for (int j = 0; j < NT; ++j) abl_b[kk][j] = bf[j];
        }
        if (NS2VC_G4_FLAG(16)) {
#pragma unroll
          for (int i = 0; i < MT; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(bf[j]));
          continue;
        }
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) MmaT<TM>::mma(acc[i][j], af[i], bf[j]);
      }
    };
    // The refill target (the stage of tile kt-1) is free for everyone after the barrier.
    // (Running the two K halves in opposite order -- one refills first, the other multiplies first -- is 1 % faster for
    // an isolated launch but 0.5 % slower inside the captured step (same-box A/B), so both refill first.)
    refill();
    multiply();
    if (++stage == STAGES) stage = 0;
  }

  if (SPEC != 0 && ewave < 0) return;             // loaders beyond the epilogue's eight waves
  gemm4_epilogue<TM, BM, LNC>(g, acc, smem, m0, n0, tid - EOFF * 64, tr, lnraw);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// LDS = max(STAGES-deep operand ring, the epilogue's per-wave transpose tiles)
static constexpr size_t gemm_lds_bytes(int bm, int bn, int stages) {
  const size_t ring = (size_t)stages * (bm + bn) * TROW;
  const size_t epi = (size_t)4 * (bm / 2) * (bn / 2 + 4) * 4;
  return ring > epi ? ring : epi;
}

static int g_gemm_flags = 0;
template <typename TM, int BM, int BN, int STAGES>
static hipError_t launch_cfg(const GemmArgs& g, hipStream_t s) {
  const int nb = (g.N / BN) * ((g.M + BM - 1) / BM);
  const size_t lds = gemm_lds_bytes(BM, BN, STAGES);
  if (g.ln_stats) hipLaunchKernelGGL((gemm2_kernel<TM, BM, BN, STAGES, true>), dim3(nb), dim3(256), lds, s, g, g_gemm_flags);
  else hipLaunchKernelGGL((gemm2_kernel<TM, BM, BN, STAGES, false>), dim3(nb), dim3(256), lds, s, g, g_gemm_flags);
  return hipGetLastError();
}

void set_gemm_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &p, sizeof(p)); }

static constexpr size_t gemm4_lds_bytes(int bm, int bn, int stages) {
  const size_t ring = (size_t)stages * (bm + bn) * TROW;
  const size_t epi = (size_t)8 * 32 * (bn / 2 + 4) * 4;
  return ring > epi ? ring : epi;
}
template <typename TM, int BM, int BN, int STAGES, int SPEC = 0>
static hipError_t launch_cfg4(const GemmArgs& g, hipStream_t s) {
  int nb = (g.N / BN) * ((g.M + BM - 1) / BM);
  if (g.gnp_x && g.gnp_sync && g.N / BN > 1) nb = 8 * (((g.M + BM - 1) / BM + 7) / 8) * (g.N / BN);   // cooperative prologue: row blocks per XCD, padded
#if NS2VC_GEMM_ABLATE
  if (g.gnp_x) hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, false, SPEC, true>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g, g_gemm_flags);
  else if (g.ln_stats) hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, true, SPEC>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g, g_gemm_flags);
  else hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, false, SPEC>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g, g_gemm_flags);
#else
  if (g.gnp_x) hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, false, SPEC, true>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g);   // (never with a LayerNorm consumer epilogue: launch_gemm checks)
  else if (g.ln_stats) hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, true, SPEC>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g);
  else hipLaunchKernelGGL((gemm4_kernel<TM, BM, BN, STAGES, false, SPEC>), dim3(nb), dim3(64 * G4Waves<SPEC>::NW), gemm4_lds_bytes(BM, BN, STAGES), s, g);
#endif
  return hipGetLastError();
}

// which of the argument checks below refused a launch (the engine and ns2vc_k_gemm append it to their error text)
static thread_local int g_gemm_fail_line = 0;
static hipError_t gemm_invalid(int line) { g_gemm_fail_line = line; return hipErrorInvalidValue; }
int last_gemm_refusal_line() { const int l = g_gemm_fail_line; g_gemm_fail_line = 0; return l; }

static int g_force_bm = 0, g_force_bn = 0, g_force_st = 0;
static int g_ts = 1, g_ts_nl = 0, g_ts_ks = -1;   // tap-sharing conv kernel on / off (tests / tuning: ns2vc_debug_set_gemm_tile(-3, 0, 0) off, (-4, 0, nl) on with nl loader waves, 0 = default)
static int g_spec = 1;   // loader / consumer tiles where the heuristic wants them; tests / tuning: ns2vc_debug_set_gemm_tile(-1, 0, 0) selects the round-2 (plain) tile choice, (-2, 0, 0) restores
void set_forced_gemm_tile(int bm, int bn, int stages) {
  if (bm == -1 || bm == -2) { g_spec = bm == -2 ? 1 : 0; return; }
  if (bm == -3 || bm == -4) { g_ts = bm == -4 ? 1 : 0; g_ts_nl = (bm == -4 && (stages == 4 || stages == 8)) ? stages : 0; g_ts_ks = bm == -4 ? (bn == 1 ? 1 : bn == 2 ? 0 : -1) : -1; return; } g_force_bm = bm; g_force_bn = bn; g_force_st = stages & 255; g_gemm_flags = stages >> 8; }

// Tile choice.  `st` 2..4 = gemm2_kernel with that ring depth; 12 / 13 = gemm4_kernel (8 waves, K split) with ring 2 / 3.
// The compiled set is exactly what this function can return:
//   gemm4: {128, 64} x 128, ring {2, 3};   gemm2 (4 waves): 64x128 ring 2 (narrow GEGLU), 64x64 ring {2, 3, 4} (N not a multiple of 128).
template <typename TM>
static hipError_t launch_typed(const GemmArgs& g, hipStream_t s) {
  const int bke = 128 / (int)sizeof(TM);
  const int nk = g.K / bke;
  int bm, bn, st;
  const bool n128 = (g.N % 128) == 0;
  if (g_force_bm) {   // test / tuning hook (ns2vc_debug_set_gemm_tile)
    bm = g_force_bm; bn = g_force_bn; st = g_force_st ? g_force_st : 3;
    if (g.N % bn || ((g.geglu || g.rowstats) && bn != 128)) return gemm_invalid(__LINE__);
  } else {
    // Tuned on MI355X with tools/gemm_sweep.py over the 10 s x batch-32 plan (profiles/gemm_sweep_r01d_bufferdma.txt).
    // The 8-wave K-split kernel wins everywhere except the narrowest GEGLU; 128-row tiles pay off once K is long
    // (>= 12 tiles) or N is wide, and only while the grid still covers the chip (M >= ~7000 rows).
    const bool big_m = g.M >= 7000;
    if (g.geglu) {
      if (!n128) return gemm_invalid(__LINE__);
      if (g.N >= 2048) { bm = 128; bn = 128; st = 12; }
      else { bm = 64; bn = 128; st = 2; }
    } else if (n128 && g.N <= 512) {
      // narrow outputs: every conv and the to_out / proj / ff-out linears (and all LayerNorm-statistics producers)
      bn = 128; st = 13;
      bm = (big_m && nk >= 12) ? 128 : 64;
      // loader / consumer waves where the tile was 64 rows anyway, and instead of the 128-row tiles of the coarse levels (one
      // workgroup per CU either way; per-launch table in profiles/r03_gemm_spec.txt)
      if (g_spec && nk >= 4 && (bm == 64 || g.M < 12000)) { bm = 64; st = 23; }
      else if (g_spec && bm == 128) st = 23;                               // ... and the 128-row tiles of levels 0-1 keep their shape, specialised (-0.8 % family, same-box)
      if (g.M >= 12000 && g.N > 128 && nk <= 2) { bm = 128; st = 12; }     // level-0 q|k|v: short K, wide-ish N
    } else if (n128) {
      bn = 128;
      if (g.M >= 12000) { bm = 128; st = 12; } else { bm = 64; st = (g_spec && nk >= 4) ? 23 : 13; }
    } else {
      bm = 64; bn = 64; st = nk >= 32 ? 4 : (nk >= 20 ? 3 : 2);
    }
  }
  if (g.gnp_x && !((st >= 12 && st <= 13) || (st >= 22 && st <= 44))) return gemm_invalid(__LINE__);   // the prologue lives in gemm4_kernel
  if (st >= 22 && st <= 44) {   // loader / consumer specialised kernels: st = 10 * (1 + SPEC) + ring depth
    if (bn != 128) return gemm_invalid(__LINE__);
#define NS2VC_CASE4S(BM_, ST_, SP_) if (bm == BM_ && st == 10 * (1 + SP_) + ST_) return launch_cfg4<TM, BM_, 128, ST_, SP_>(g, s)
    // compiled: 4 + 4 waves, ring 3.  Measured and not kept (profiles/r03_gemm_spec.txt): 8 + 8 and 8 + 4 waves (no faster at one
    // workgroup per CU, slower at two: 1024 / 768 threads), ring 2 (+11 % in the captured step) and ring 4 (one workgroup per CU)
    NS2VC_CASE4S(128, 3, 1); NS2VC_CASE4S(64, 3, 1);
#undef NS2VC_CASE4S
    return gemm_invalid(__LINE__);
  }
  if (st == 12 || st == 13) {   // 8-wave K-split kernel, ring depth st - 10
    if (bn != 128) return gemm_invalid(__LINE__);
#define NS2VC_CASE4(BM_, ST_) if (bm == BM_ && st == 10 + ST_) return launch_cfg4<TM, BM_, 128, ST_>(g, s)
    NS2VC_CASE4(128, 2); NS2VC_CASE4(128, 3); NS2VC_CASE4(64, 2); NS2VC_CASE4(64, 3);
#undef NS2VC_CASE4
    return gemm_invalid(__LINE__);
  }
#define NS2VC_CASE(BM_, BN_, ST_) if (bm == BM_ && bn == BN_ && st == ST_) return launch_cfg<TM, BM_, BN_, ST_>(g, s)
  NS2VC_CASE(64, 128, 2);
  NS2VC_CASE(64, 64, 2); NS2VC_CASE(64, 64, 3); NS2VC_CASE(64, 64, 4);
#undef NS2VC_CASE
  return gemm_invalid(__LINE__);
}

hipError_t launch_gemm(const GemmArgs& g_, int prec, hipStream_t s) {
  GemmArgs g = g_;
  if (g.N % 64 != 0 || g.M <= 0) return gemm_invalid(__LINE__);
  if (g.N / 64 > 15) g.gnp_sync = nullptr;          // the arrival word counts the sharers of a row block per XCC in 4 bits (gnpro.h)
  const int bke = prec == PREC_F32 ? 32 : 64;
  if (g.K % bke != 0 || g.c0 % bke != 0 || g.c1 % bke != 0 || g.c2 % bke != 0 || g.K != g.taps * (g.c0 + g.c1) + g.c2) return gemm_invalid(__LINE__);
  if (g.c2 && (!g.a2 || g.tmode != TMODE_SAME || (g.lda2 % (bke / 8)))) return gemm_invalid(__LINE__);
  if ((g.lda0 % (bke / 8)) || (g.c1 && (g.lda1 % (bke / 8)))) return gemm_invalid(__LINE__);    // 16-B aligned rows
  if (!g.out_f32 && !g.out_op && !g.sol_coef) return gemm_invalid(__LINE__);     // (a solver epilogue's outputs are the state tensors)
  if (g.stats && (g.geglu || g.Tout < 64 || (g.N & 15))) return gemm_invalid(__LINE__);
  if (g.rowstats && g.geglu) return gemm_invalid(__LINE__);
  if (g.ln_stats && (!g.ln_wsum || g.ln_dim <= 0 || (g.ln_dim & 127) || g.ln_dim > 512)) return gemm_invalid(__LINE__);
  if (g.rowstats && (g.N & 127)) return gemm_invalid(__LINE__);
  if ((g.out_f32 && (g.ldo_f32 & 3)) || (g.out_op && (g.ldo_op & 3)) || (g.res && (g.ldres & 3))) return gemm_invalid(__LINE__);   // 16-B row segments
  if (g.gnp_x) {     // GroupNorm-apply prologue: one or two (concatenated) sources, same-length rows, whole 16-channel blocks per group, <= 3 batch items per tile + halo
    // (gnp_pair: a0 holds the hi + lo planes of c0 / 2 normalised channels; the hi plane may be read once more through a1 = a0)
    const int cn = g.gnp_pair ? g.c0 >> 1 : g.c0;
    if (g.gnp_pair ? (prec == PREC_F32 || (g.c0 & 7) || (g.c1 && (g.a1 != g.a0 || g.c1 != cn || g.lda1 != g.lda0))) : g.c1 != 0) return gemm_invalid(__LINE__);
    if (g.ln_stats || g.tmode != TMODE_SAME || g.Tin != g.Tout || g.geglu || (g.N & 127) || g.c0 > 1024 || (cn & 3) || !g.gnp_stats || !g.gnp_gamma ||
        !g.gnp_beta || g.gnp_G < 1 || g.gnp_G > 8 || (cn % g.gnp_G) || ((cn / g.gnp_G) & 15) || g.Tin < 66 || (g.gnp_ldx & 3) || (g.lda0 & 3))
      return gemm_invalid(__LINE__);
    // cooperative form: rows written by other CUs are read back without an L1 invalidate, which is only sound while no cache line holds
    // rows of two row blocks -- whole 128-byte lines per row
    if (g.gnp_sync && ((reinterpret_cast<uintptr_t>(g.a0) & 127) || (((size_t)g.lda0 * operand_bytes(prec)) & 127) || (reinterpret_cast<uintptr_t>(g.gnp_sync) & 7)))
      return gemm_invalid(__LINE__);
    if (g.gnp_c1 && (g.gnp_c1 < 0 || g.gnp_c1 >= cn || (g.gnp_c1 & 15) || ((cn - g.gnp_c1) & 15) || !g.gnp_x1 || !g.gnp_stats1 || (g.gnp_ldx1 & 3)))
      return gemm_invalid(__LINE__);                         // a concat of two sources: whole 16-channel blocks from each
  }
  {   // the DMA addresses rows by 32-bit byte offsets from each tensor's base: every operand must stay below 4 GB
    const unsigned long long sz = operand_bytes(prec), lim = 0xFFF00000ull;
    const unsigned long long rows = (unsigned long long)g.B * g.Tin;
    if (rows * g.lda0 * sz > lim || (g.c1 && rows * g.lda1 * sz > lim) || (g.c2 && rows * g.lda2 * sz > lim) ||
        (unsigned long long)g.N * g.K * sz > lim)
      return gemm_invalid(__LINE__);
  }
  // k = 3 / stride 1: the tap-sharing kernel (convts.hip) unless the caller (algo = 1), the global switch or a forced gemm4 / gemm2 tile says otherwise;
  // a forced tile (128, 64 | 128, 54 | 58) selects its BN and loader-wave count
  {
    // (stages 64 | 68: the same with the K-split consumer layout, 64-column tiles only)
    const bool forced_ks = g_force_bm == 128 && g_force_bn == 64 && (g_force_st == 64 || g_force_st == 68);
    const bool forced_ts = (g_force_bm == 128 && (g_force_st == 54 || g_force_st == 58)) || forced_ks;
    if (g.algo != 1 && (forced_ts || (g_ts && !g_force_bm)) && convts_eligible(g, prec))
      return launch_convts(g, prec, forced_ts ? g_force_bn : 0, forced_ts ? g_force_st % 10 : g_ts_nl, forced_ts ? (forced_ks ? 1 : 0) : g_ts_ks, s);
    if (forced_ts) return gemm_invalid(__LINE__);
  }
  if (g.sol_coef) return gemm_invalid(__LINE__);             // the solver epilogue exists in the tap-sharing kernel only
  if (g.gnp_pair) return gemm_invalid(__LINE__);             // ... and so does the prologue that writes hi + lo operand pairs
  switch (prec) {
    case PREC_BF16: return launch_typed<bf16_t>(g, s);
    case PREC_F16: return launch_typed<f16_t>(g, s);
    case PREC_F32: return launch_typed<float>(g, s);
    default: return gemm_invalid(__LINE__);
  }
}

// would launch_gemm hand this launch to the tap-sharing conv kernel right now?  (the engine asks before it folds the solver update into conv_out)
bool gemm_uses_convts(const GemmArgs& g, int prec) {
  const bool forced_ks = g_force_bm == 128 && g_force_bn == 64 && (g_force_st == 64 || g_force_st == 68);
  const bool forced_ts = (g_force_bm == 128 && (g_force_st == 54 || g_force_st == 58)) || forced_ks;
  return g.algo != 1 && (forced_ts || (g_ts && !g_force_bm)) && convts_eligible(g, prec);
}

template <typename K> static hipError_t set_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

#define NS2VC_SET(TM, BM, BN, ST)                                                                              \
  do {                                                                                                          \
    hipError_t e = set_lds(gemm2_kernel<TM, BM, BN, ST, false>, gemm_lds_bytes(BM, BN, ST));                    \
    if (e == hipSuccess) e = set_lds(gemm2_kernel<TM, BM, BN, ST, true>, gemm_lds_bytes(BM, BN, ST));           \
    if (e != hipSuccess) return e;                                                                              \
  } while (0)
#define NS2VC_SET4(TM, BM, ST)                                                                                  \
  do {                                                                                                          \
    hipError_t e = set_lds(gemm4_kernel<TM, BM, 128, ST, false>, gemm4_lds_bytes(BM, 128, ST));                 \
    if (e == hipSuccess) e = set_lds(gemm4_kernel<TM, BM, 128, ST, true>, gemm4_lds_bytes(BM, 128, ST));        \
    if (e == hipSuccess) e = set_lds(gemm4_kernel<TM, BM, 128, ST, false, 0, true>, gemm4_lds_bytes(BM, 128, ST)); \
    if (e != hipSuccess) return e;                                                                              \
  } while (0)
#define NS2VC_SET4S(TM, BM, ST, SP)                                                                             \
  do {                                                                                                          \
    hipError_t e = set_lds(gemm4_kernel<TM, BM, 128, ST, false, SP>, gemm4_lds_bytes(BM, 128, ST));             \
    if (e == hipSuccess) e = set_lds(gemm4_kernel<TM, BM, 128, ST, true, SP>, gemm4_lds_bytes(BM, 128, ST));    \
    if (e == hipSuccess) e = set_lds(gemm4_kernel<TM, BM, 128, ST, false, SP, true>, gemm4_lds_bytes(BM, 128, ST)); \
    if (e != hipSuccess) return e;                                                                              \
  } while (0)
template <typename TM> static hipError_t init_typed() {
  NS2VC_SET4(TM, 128, 2); NS2VC_SET4(TM, 128, 3); NS2VC_SET4(TM, 64, 2); NS2VC_SET4(TM, 64, 3);
  NS2VC_SET4S(TM, 128, 3, 1); NS2VC_SET4S(TM, 64, 3, 1);
  NS2VC_SET(TM, 64, 128, 2);
  NS2VC_SET(TM, 64, 64, 2); NS2VC_SET(TM, 64, 64, 3); NS2VC_SET(TM, 64, 64, 4);
  return hipSuccess;
}
hipError_t init_gemm_attributes() {
  hipError_t e = init_typed<float>();
  if (e == hipSuccess) e = init_typed<bf16_t>();
  if (e == hipSuccess) e = init_typed<f16_t>();
  return e;
}

}  // namespace ns2vc
