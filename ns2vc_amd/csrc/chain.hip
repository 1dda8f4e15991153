// Fused row-panel chains for the transformer blocks of the NS2VC denoiser (gfx950, bf16 operand type).
//
// The per-token (row-local) part of BasicTransformerBlock / Transformer2DModel (reference attention.py:130-203,
// transformer_1d.py:256-295) is a chain of small Linears (N, K in {128..512 / 4D}) separated by LayerNorms and
// residual adds.  As separate GEMM launches each CU gets ~one 128x128 tile of work per launch and the launches are
// L2-bandwidth / fill-drain bound (profiles/gemm_sweep_r01.txt).  Here one workgroup owns BM rows for the WHOLE
// chain: the activation panel [BM][D] lives in LDS (A operand of every MFMA), intermediate results never leave the
// chip, and the only stream from L2 is the weights — pre-packed on the host as a flat sequence of 16 KB tiles
// ([128 rows][64 k] bf16, already in the bank-conflict-free XOR-swizzled LDS image, in exactly the order the kernel
// consumes them), so the loader is a lane-linear global_load_lds into an S-deep ring with counted vmcnt waits.
//
//   chain_ab_kernel :  y = A*W1^T + b1 (+ res)  ->  n = LayerNorm(y) (affine folded into W2)  ->  out2 = n*W2^T + b2
//                      (proj_in -> norm1 -> q|k|v   and   attn1.to_out + residual -> norm2 -> attn2.to_q)
//   chain_ff_kernel :  y1 = A*Wo^T + bo + y -> n = LN(y1) -> GEGLU(n*W1^T + b1) * W2^T + b2 + y1 -> * Wp^T + bp + x
//                      (attn2.to_out + residual -> norm3 -> feed-forward -> + residual -> proj_out + block residual)
//
// 256 threads = 4 waves; wave tile 32 x (128 / WGN) of each 128-column output chunk, 32x32x16 bf16 MFMA.
#include "common.h"
#include "mma.h"
#include <cstdlib>

namespace ns2vc {

constexpr int TILE_BYTES = 128 * 128;      // one weight tile: 128 rows x 128 B of K

template <int N> struct WaitSel {
  __device__ static __forceinline__ void wait(int after) {     // after = number of later tiles allowed in flight
    switch (after) {
      case 0: wait_vmcnt<0>(); break;
      case 1: wait_vmcnt<4>(); break;
      case 2: wait_vmcnt<8>(); break;
      case 3: wait_vmcnt<12>(); break;
      case 4: wait_vmcnt<16>(); break;
      case 5: wait_vmcnt<20>(); break;
      default: wait_vmcnt<24>(); break;
    }
  }
};

// Everything the two chain kernels share: LDS carve-up, the weight stream, the panel MFMA step and the
// per-chunk transposed epilogue iteration.
template <int BM, int S>
struct Chain {
  static constexpr int WGM = BM / 32, WGN = 4 / WGM, WN = 128 / WGN, NT = WN / 32;
  static constexpr int EP = WN + 4;                 // epilogue staging pitch (floats)
  static constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = 32 / RPI;
  static constexpr int SLOTS = 16;                  // row-statistic slots per row: (D/128 chunks) x WGN <= 16

  char* panel; char* ring; float* stage; float* rowstat; float* rowmr;
  unsigned panel_lds, ring_lds;
  int tid, lane, wave, wm, wn, l31, hi, sw, rsub, cq;
  const char* wstream;
  int t_issue, total;

  __device__ __forceinline__ void init(char* smem, int D, const void* ws, int total_tiles) {
    tid = threadIdx.x; lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    wm = wave / WGN; wn = wave % WGN;
    l31 = lane & 31; hi = lane >> 5; sw = (l31 >> 1) & 7;
    rsub = lane / LPR; cq = lane % LPR;
    panel = smem;
    ring = panel + (size_t)BM * D * 2;
    stage = reinterpret_cast<float*>(ring + S * TILE_BYTES) + wave * (32 * EP);
    rowstat = reinterpret_cast<float*>(ring + S * TILE_BYTES) + 4 * 32 * EP;
    rowmr = rowstat + BM * SLOTS * 2;
    panel_lds = (unsigned)(size_t)panel; ring_lds = (unsigned)(size_t)ring;
    wstream = reinterpret_cast<const char*>(ws);
    t_issue = 0; total = total_tiles;
  }
  static size_t lds_bytes(int D) { return (size_t)BM * D * 2 + (size_t)S * TILE_BYTES + (size_t)4 * 32 * EP * 4 + (size_t)BM * SLOTS * 2 * 4 + (size_t)BM * 2 * 4; }

  // operand panel [BM][D] bf16 -> LDS as D/64 swizzled k-tiles (source-side swizzle, rows >= M read zeros)
  __device__ __forceinline__ void load_panel(const void* a, int M, int D, int m0) {
    const unsigned long long zero = reinterpret_cast<unsigned long long>(g_zero_page);
    const int npass = BM * D / 8 / 256;
    for (int j = 0; j < npass; ++j) {
      const int i = tid + 256 * j;
      const int kt = i / (BM * 8), rem = i - kt * (BM * 8);
      const int row = rem >> 3, pc = rem & 7;
      const int lc = pc ^ ((row >> 1) & 7);
      const int m = m0 + row;
      const unsigned long long src = reinterpret_cast<unsigned long long>(a) + ((size_t)m * D + kt * 64 + lc * 8) * 2;
      glds16(reinterpret_cast<const void*>(m < M ? src : zero), panel_lds + (256 * j + wave * 64) * 16);
    }
  }
  __device__ __forceinline__ void issue_w() {         // next tile of the weight stream -> ring
    if (t_issue < total) {
      const char* src = wstream + (size_t)t_issue * TILE_BYTES + tid * 16;
      const unsigned dst = ring_lds + (t_issue % S) * TILE_BYTES + wave * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j) glds16(src + j * 4096, dst + j * 4096);
    }
    ++t_issue;
  }
  // wait until weight tile t (and everything issued before it) has landed, publish it, refill the ring
  __device__ __forceinline__ void acquire(int t) {
    WaitSel<S>::wait(min(S - 2, total - 1 - t));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_w();
  }
  // acc[j] += panel_tile(kt) [wave rows] x ring_tile(t) [wave cols]
  __device__ __forceinline__ void mma_tile(f32x16_t (&acc)[NT], const char* ptile, int t) {
    const char* ap = ptile + (wm * 32 + l31) * 128;
    const char* bp = ring + (t % S) * TILE_BYTES + (wn * WN + l31) * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + hi) ^ sw) * 16;
      const u32x4_t af = *reinterpret_cast<const u32x4_t*>(ap + coff);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const u32x4_t bf = *reinterpret_cast<const u32x4_t*>(bp + j * 32 * 128 + coff);
        MmaT<bf16_t>::mma(acc[j], af, bf);
      }
    }
  }
  // C layout -> wave-private staging tile (row-major), so the epilogue can move whole rows
  __device__ __forceinline__ void stage_acc(const f32x16_t (&acc)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[(8 * (r >> 2) + 4 * hi + (r & 3)) * EP + j * 32 + l31] = acc[j][r];
  }
  __device__ __forceinline__ float4 staged(int it) const { return *reinterpret_cast<const float4*>(stage + (it * RPI + rsub) * EP + cq * 4); }
  // block row / chunk-local column of epilogue iteration `it`
  __device__ __forceinline__ int erow(int it) const { return wm * 32 + it * RPI + rsub; }
  __device__ __forceinline__ int ecol() const { return wn * WN + cq * 4; }
  // bf16x4 -> panel position (row r, column c of the [BM][D] panel)
  __device__ __forceinline__ void panel_store4(int r, int c, float a, float b, float c2, float d) {
    char* p = panel + (c >> 6) * (BM * 128) + r * 128 + ((((c & 63) >> 3) ^ ((r >> 1) & 7)) << 4) + ((c & 7) >> 2) * 8;
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c2, d));
  }
};

struct ChainABArgs {
  const void* a; int M, D;
  const void* wstream;
  const float* bias1; const float* res; float* y; float eps;
  const float* bias2; void* out2; int N2;
};

template <int BM, int S>
__global__ __launch_bounds__(256) void chain_ab_kernel(const ChainABArgs g) {
  using CH = Chain<BM, S>;
  constexpr int NT = CH::NT, NIT = CH::NIT, LPR = CH::LPR, SLOTS = CH::SLOTS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  CH c;
  const int D = g.D, nkD = D >> 6, nc1 = D >> 7, nc2 = g.N2 >> 7;
  c.init(smem, D, g.wstream, (nc1 + nc2) * nkD);
  const int m0 = blockIdx.x * BM;
  c.load_panel(g.a, g.M, D, m0);
#pragma unroll
  for (int s = 0; s < S - 1; ++s) c.issue_w();

  int t = 0;
  // ---------------- phase 1: y = A*W1^T + b1 (+res), row statistics ----------------
  for (int nc = 0; nc < nc1; ++nc) {
    f32x16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kt = 0; kt < nkD; ++kt, ++t) {
      c.acquire(t);
      c.mma_tile(acc, c.panel + kt * (BM * 128), t);
    }
    c.stage_acc(acc);
    __syncthreads();
    const int col = nc * 128 + c.ecol();
    const float4 bv = *reinterpret_cast<const float4*>(g.bias1 + col);
    float4 rr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + c.erow(it);
      rr[it] = (g.res && m < g.M) ? *reinterpret_cast<const float4*>(g.res + (size_t)m * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = c.erow(it), m = m0 + row;
      const float4 a4 = c.staged(it);
      float4 v;
      v.x = a4.x + bv.x + rr[it].x; v.y = a4.y + bv.y + rr[it].y; v.z = a4.z + bv.z + rr[it].z; v.w = a4.w + bv.w + rr[it].w;
      float ps = 0.f, pq = 0.f;
      if (m < g.M) {
        *reinterpret_cast<float4*>(g.y + (size_t)m * D + col) = v;
        ps = (v.x + v.y) + (v.z + v.w);
        pq = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) { ps += __shfl_xor(ps, o); pq += __shfl_xor(pq, o); }
      if (c.cq == 0) {
        float* st = c.rowstat + (row * SLOTS + nc * CH::WGN + c.wn) * 2;
        st[0] = ps; st[1] = pq;
      }
    }
    __syncthreads();
  }
  // ---------------- LayerNorm statistics (fixed summation order, double) ----------------
  if (c.tid < BM) {
    double s = 0.0, q = 0.0;
    const float* st = c.rowstat + c.tid * SLOTS * 2;
    for (int k = 0; k < nc1 * CH::WGN; ++k) { s += (double)st[2 * k]; q += (double)st[2 * k + 1]; }
    const double mean = s / (double)D;
    double var = q / (double)D - mean * mean;
    if (var < 0.0) var = 0.0;
    c.rowmr[2 * c.tid] = (float)mean;
    c.rowmr[2 * c.tid + 1] = (float)(1.0 / sqrt(var + (double)g.eps));
  }
  __threadfence_block();
  __syncthreads();
  // ---------------- normalised panel: every thread re-reads the y values it stored ----------------
  for (int nc = 0; nc < nc1; ++nc) {
    const int col = nc * 128 + c.ecol();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = c.erow(it), m = m0 + row;
      const float mu = c.rowmr[2 * row], rs = c.rowmr[2 * row + 1];
      float4 v = make_float4(mu, mu, mu, mu);
      if (m < g.M) v = *reinterpret_cast<const float4*>(g.y + (size_t)m * D + col);
      c.panel_store4(row, col, (v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
    }
  }
  __syncthreads();
  // ---------------- phase 2: out2 = n*W2^T + b2 (operand-typed) ----------------
  bf16_t* o2 = reinterpret_cast<bf16_t*>(g.out2);
  for (int nc = 0; nc < nc2; ++nc) {
    f32x16_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int kt = 0; kt < nkD; ++kt, ++t) {
      c.acquire(t);
      c.mma_tile(acc, c.panel + kt * (BM * 128), t);
    }
    c.stage_acc(acc);
    __syncthreads();
    const int col = nc * 128 + c.ecol();
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias2) bv = *reinterpret_cast<const float4*>(g.bias2 + col);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + c.erow(it);
      const float4 a4 = c.staged(it);
      if (m < g.M) store_op4<bf16_t>(o2 + (size_t)m * g.N2 + col, a4.x + bv.x, a4.y + bv.y, a4.z + bv.z, a4.w + bv.w);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int BM, int S> static hipError_t launch_ab_t(const ChainABArgs& g, hipStream_t s) {
  const size_t lds = Chain<BM, S>::lds_bytes(g.D);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_ab_kernel<BM, S>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((chain_ab_kernel<BM, S>), dim3((g.M + BM - 1) / BM), dim3(256), lds, s, g);
  return hipGetLastError();
}

hipError_t launch_chain_ab(const void* a, int M, int D, const void* wstream, const float* bias1, const float* res, float* y, float eps,
                           const float* bias2, void* out2, int N2, hipStream_t s) {
  if ((D & 127) || (N2 & 127) || D > 512 || M <= 0) return hipErrorInvalidValue;
  ChainABArgs g;
  g.a = a; g.M = M; g.D = D; g.wstream = wstream; g.bias1 = bias1; g.res = res; g.y = y; g.eps = eps;
  g.bias2 = bias2; g.out2 = out2; g.N2 = N2;
  // 32-row panels with a shallow ring keep the LDS footprint under 80 KB so TWO workgroups share a CU: the chain is a
  // sequence of dependent global round trips (panel, residual, weights, stores) and a second resident workgroup is
  // what hides them.  (Measured on MI355X: 64-row panels at one workgroup per CU were slower at every level.)
  static const int cfg = getenv("NS2VC_CHAIN_CFG") ? atoi(getenv("NS2VC_CHAIN_CFG")) : 0;
  if (cfg == 1) return launch_ab_t<64, 5>(g, s);
  if (cfg == 2) return launch_ab_t<32, 6>(g, s);
  if (cfg == 3) return launch_ab_t<32, 3>(g, s);
  if (cfg == 4) return launch_ab_t<64, 2>(g, s);
  if (cfg == 5) return launch_ab_t<64, 3>(g, s);
  return launch_ab_t<32, 2>(g, s);
}

}  // namespace ns2vc
