// conv1 (k = 3) of a ResnetBlock + GroupNorm(norm2) + time scale / shift + SiLU in ONE launch, CDNA4 (gfx950), 16-bit operands.
//
// Reference: unet1d/resnet.py:591-641 --  h = conv1(act(norm1(x)));  h = norm2(h) * (1 + scale) + shift;  h = act(h)  -- the
// middle of every ResnetBlock2D.  As two launches (round 1-2) the implicit GEMM wrote h as an fp32 tensor plus its GroupNorm
// statistics and gn_apply_kernel read it back to write the operand tensor the second convolution consumes: 46 MB of HBM
// traffic and two dependent kernels per site at level 0.  GroupNorm needs the statistics of a whole (batch item, group)
// before a single element can be normalised, which is what kept it out of the producer: a workgroup owned a 64-row tile.
// Here a workgroup owns a whole REDUCTION DOMAIN instead -- all T frames of one batch item for a slice of NS = lcm(32, C/G)
// output channels (whole groups) -- so the statistics are complete inside the workgroup, the convolution result never
// leaves the registers, and what is written is the finished operand tensor (the fp32 h and the gn_apply launch disappear).
//
//   * 8 waves; wave w owns frames [w * 32 RT, (w+1) * 32 RT) of the item (RT = ceil(T / 256) <= 4) for all NS channels:
//     every activation row is needed by exactly one wave, so there is nothing to share through LDS and no barrier in the
//     K loop: both operands go global -> VGPR (buffer loads; taps / zero padding / frames past T are out-of-range offsets
//     that read as zeros), double-buffered in registers one half K tile (32 k) ahead;
//   * computed TRANSPOSED, out^T[channel][frame] = W a^T (as rowchain.hip / ffn.hip): a lane owns one frame and 16
//     channels per 32-channel block, so bias, affine, SiLU and the 16-byte operand stores are per-lane arithmetic and the
//     GroupNorm sums are a per-lane sum followed by one wave reduction and a fixed-order sum over the eight waves;
//   * weights are packed on the host in fragment order (one coalesced 1-KB load per 32 x 16 fragment).
#include "common.h"
#include <cstdlib>
#include <vector>

namespace ns2vc {

typedef ::ns2vc_convgn_args ConvGnArgs;
typedef _Float16 cg_f16x8_t __attribute__((ext_vector_type(8)));

template <typename TM> struct CgMma;
template <> struct CgMma<bf16_t> {
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    union U { u32x4_t u; bf16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
  }
};
template <> struct CgMma<f16_t> {
  __device__ static __forceinline__ void mma(f32x16_t& acc, const u32x4_t& a, const u32x4_t& b) {
    union U { u32x4_t u; cg_f16x8_t v; };
    U ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.v, ub.v, acc, 0, 0, 0);
  }
};

constexpr unsigned CG_OOB = 0xFFFFFFF0u;     // voffset past every descriptor's range: the load returns zeros

template <typename TM, int RT, int CT>
__global__ __launch_bounds__(512) void convgn_kernel(const ConvGnArgs a) {
  op_mode_init<TM>();
  constexpr int NS = 32 * CT;                 // channels of this workgroup
  constexpr int NBLK = 2 * CT;                // 16-channel blocks of the slice
  __shared__ float red[8][NBLK][2];           // per wave and block: (sum, sum of squares)
  __shared__ float2 gstat[NBLK];              // per block: (mean, rstd) of the group it belongs to

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int NSL = a.N / NS;
  // XCD-aware mapping: the slices of one batch item read the same activation rows -> consecutive ids on one XCD
  int b, sl;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    b = swz / NSL;
    sl = swz - b * NSL;
  }
  const int n0 = sl * NS;
  const int KT = a.cin >> 6;                  // 64-wide K tiles per tap
  const int T = a.T;

  const unsigned long long abytes = (unsigned long long)a.B * T * a.lda * 2ull;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.a), 0, (int)(abytes > 0xFFFFFFE0ull ? 0xFFFFFFE0ull : abytes), 0x00020000);
  const unsigned wslice = (unsigned)(3 * KT) * 4u * CT * 1024u;          // bytes of one slice's fragment stream
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.wpack)) + (size_t)sl * wslice, 0, (int)wslice, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;

  // per-lane byte offsets of the three tap rows of every frame this lane owns (frame t, taps t-1, t, t+1)
  unsigned voff[RT][3];
  bool tok_ok[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int t = wave * 32 * RT + 32 * rt + l31;
    tok_ok[rt] = t < T;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int tp = t + tap - 1;
      const bool ok = t < T && tp >= 0 && tp < T;
      voff[rt][tap] = ok ? (unsigned)(b * T + tp) * (unsigned)a.lda * 2u + 16u * (unsigned)hi : CG_OOB;
    }
  }

  f32x16_t acc[CT][RT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][rt][r] = 0.f;

  // one half K tile = two 16-wide k slabs: activation fragments fa[rt][j], weight fragments fw[ct][j]
  struct Frag { u32x4_t fa[RT][2]; u32x4_t fw[CT][2]; };
  auto load_half = [&](Frag& f, const unsigned (&vo)[RT], int kt, int ktin, int half) __attribute__((always_inline)) {
    const int soffA = (ktin * 64 + half * 32) * 2;                       // bytes inside the row (+ 16 hi in the lane offset)
    const int soffW = ((kt * 4 + half * 2) * CT) * 1024;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int j = 0; j < 2; ++j) f.fw[ct][j] = __builtin_amdgcn_raw_buffer_load_b128(rW, lane16, soffW + (j * CT + ct) * 1024, 0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int j = 0; j < 2; ++j) f.fa[rt][j] = __builtin_amdgcn_raw_buffer_load_b128(rA, vo[rt], soffA + 32 * j, 0);
  };
  auto mma_half = [&](const Frag& f) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) CgMma<TM>::mma(acc[ct][rt], f.fw[ct][j], f.fa[rt][j]);
  };
  unsigned vo0[RT], vo1[RT], vo2[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { vo0[rt] = voff[rt][0]; vo1[rt] = voff[rt][1]; vo2[rt] = voff[rt][2]; }

  // K loop: tap-major.  Within a tap: [half 0 of tile ktin in fA] -> load half 1 -> MFMAs(fA) -> load half 0 of the next tile -> MFMAs(fB)
  Frag fA, fB;
  load_half(fA, vo0, 0, 0, 0);
  // (scheduling fences: left alone the compiler sinks every load to just in front of the MFMA that uses it and re-uses the
  //  fragment registers -- `s_waitcnt vmcnt(1)` before every MFMA, no prefetch distance at all)
  auto tap_loop = [&](const unsigned (&vo)[RT], const unsigned (&vo_next)[RT], int kt_base, bool has_next) __attribute__((always_inline)) {
    for (int ktin = 0; ktin + 1 < KT; ++ktin) {
      load_half(fB, vo, kt_base + ktin, ktin, 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(fA);
      __builtin_amdgcn_sched_barrier(0);
      load_half(fA, vo, kt_base + ktin + 1, ktin + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      mma_half(fB);
      __builtin_amdgcn_sched_barrier(0);
    }
    load_half(fB, vo, kt_base + KT - 1, KT - 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_half(fA);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) load_half(fA, vo_next, kt_base + KT, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma_half(fB);
    __builtin_amdgcn_sched_barrier(0);
  };
  tap_loop(vo0, vo1, 0, true);
  tap_loop(vo1, vo2, KT, true);
  tap_loop(vo2, vo2, 2 * KT, false);

  // ---- epilogue.  Register r of acc[ct][rt] <-> channel n0 + 32 ct + 8 (r >> 2) + 4 hi + (r & 3), frame of block rt.
  float4 bq[CT][4];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[ct][q] = *reinterpret_cast<const float4*>(a.bias + n0 + 32 * ct + 8 * q + 4 * hi);
  float bs[NBLK], bss[NBLK];
#pragma unroll
  for (int k = 0; k < NBLK; ++k) { bs[k] = 0.f; bss[k] = 0.f; }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v0 = acc[ct][rt][4 * q + 0] + bq[ct][q].x, v1 = acc[ct][rt][4 * q + 1] + bq[ct][q].y;
        const float v2 = acc[ct][rt][4 * q + 2] + bq[ct][q].z, v3 = acc[ct][rt][4 * q + 3] + bq[ct][q].w;
        acc[ct][rt][4 * q + 0] = v0; acc[ct][rt][4 * q + 1] = v1; acc[ct][rt][4 * q + 2] = v2; acc[ct][rt][4 * q + 3] = v3;
        if (tok_ok[rt]) {
          bs[2 * ct + (q >> 1)] += (v0 + v1) + (v2 + v3);
          bss[2 * ct + (q >> 1)] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
        }
      }
  // every lane of the wave contributes to every block: full-wave butterfly (fixed order), then the eight waves through LDS
#pragma unroll
  for (int k = 0; k < NBLK; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bs[k] += __shfl_xor(bs[k], o); bss[k] += __shfl_xor(bss[k], o); }
    if (lane == 0) { red[wave][k][0] = bs[k]; red[wave][k][1] = bss[k]; }
  }
  __syncthreads();
  {
    const int Cg = a.N / a.G, nb = Cg >> 4;                            // 16-channel blocks per group
    if (tid < NBLK / nb) {                                              // one thread per group of the slice
      double ds = 0.0, dq = 0.0;
      for (int j = 0; j < nb; ++j)
        for (int w = 0; w < 8; ++w) { ds += (double)red[w][tid * nb + j][0]; dq += (double)red[w][tid * nb + j][1]; }
      // (the same finalisation as gn_apply_kernel, misc.hip)
      const float inv_nf = 1.0f / ((float)T * (float)Cg);
      const double inv_n = (double)inv_nf * (2.0 - (double)inv_nf * ((double)T * (double)Cg));
      const double mean = ds * inv_n;
      double var = dq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float ve = (float)var + a.eps;
      float r = rsqrtf(ve);
      r = r * (1.5f - 0.5f * ve * r * r);
      for (int j = 0; j < nb; ++j) gstat[tid * nb + j] = make_float2((float)mean, r);
    }
  }
  __syncthreads();
  if (a.dbg_stats && tid < NBLK) {                 // test hook: (mean, rstd) per (batch item, 16-channel block)
    a.dbg_stats[((size_t)b * (a.N >> 4) + (n0 >> 4) + tid) * 2 + 0] = gstat[tid].x;
    a.dbg_stats[((size_t)b * (a.N >> 4) + (n0 >> 4) + tid) * 2 + 1] = gstat[tid].y;
  }

  TM* const oo = reinterpret_cast<TM*>(a.out_op);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    // per-channel scale / shift of this lane's 16 channels of the block: y = x * sc + sh  (gn_apply_kernel's arithmetic)
    float sc[4][4], sh[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + 32 * ct + 8 * q + 4 * hi;
      const float2 mr = gstat[2 * ct + (q >> 1)];
      const float4 ga = *reinterpret_cast<const float4*>(a.gamma + n), be = *reinterpret_cast<const float4*>(a.beta + n);
      const float gam[4] = {ga.x, ga.y, ga.z, ga.w}, bet[4] = {be.x, be.y, be.z, be.w};
      float ts[4] = {0.f, 0.f, 0.f, 0.f}, tf[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.temb) {
        const float* tp = a.temb + (size_t)b * a.ldtemb + a.temb_off + n;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ts[e] = tp[e]; tf[e] = tp[a.N + e]; }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[q][e] = mr.y * gam[e];
        sh[q][e] = bet[e] - mr.x * sc[q][e];
        if (a.temb) {
          const float s1 = 1.0f + ts[e];
          sc[q][e] *= s1;
          sh[q][e] = sh[q][e] * s1 + tf[e];
        }
      }
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int t = wave * 32 * RT + 32 * rt + l31;
      uint32_t pk[4][2];
      if (a.dbg_conv && t < T) {                   // test hook: the convolution result (+ bias) before the normalisation
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(a.dbg_conv + ((size_t)b * T + t) * a.N + n0 + 32 * ct + 8 * q + 4 * hi) =
              make_float4(acc[ct][rt][4 * q], acc[ct][rt][4 * q + 1], acc[ct][rt][4 * q + 2], acc[ct][rt][4 * q + 3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y[e] = acc[ct][rt][4 * q + e] * sc[q][e] + sh[q][e];
          if (a.silu) y[e] = silu_f(y[e]);
        }
        pk[q][0] = Op16<TM>::pack(y[0], y[1]);
        pk[q][1] = Op16<TM>::pack(y[2], y[3]);
      }
      // the two lane halves trade register groups so that every lane stores 8 consecutive channels (16 B): groups (2 gp, 2 gp + 1)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const auto x0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][0], pk[2 * gp + 1][0], false, false);
        const auto x1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp][1], pk[2 * gp + 1][1], false, false);
        const int n = n0 + 32 * ct + 8 * (2 * gp + hi);
        if (t < T) out_store16(oo + ((size_t)b * T + t) * a.ldo + n, x0[0], x1[0], x0[1], x1[1]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// channels per workgroup = lcm(32, C / G) (whole GroupNorm groups, whole 32-channel MFMA blocks)
static int convgn_ct(int N, int G) {
  if (G <= 0 || N % G) return 0;
  const int Cg = N / G;
  if (Cg % 16) return 0;
  int ns = 32;
  while (ns % Cg) ns += 32;
  if (ns > 96 || N % ns) return 0;
  return ns / 32;
}

bool convgn_eligible(int cin, int N, int G, int T, int prec) {
  if (prec != PREC_BF16 && prec != PREC_F16) return false;
  if (cin <= 0 || (cin & 63) || T < 1 || (T + 255) / 256 > 4) return false;
  const int ct = convgn_ct(N, G);
  return ct != 0 && ct * ((T + 255) / 256) <= 6;      // accumulators + two fragment sets must fit 256 VGPRs without spilling
}

// rows [N][3 * cin] (K = tap * cin + c, as every conv is packed) -> fragment stream: [slice][k tile][slab][block][lane][8]
hipError_t pack_convgn_stream(const float* rows, int N, int cin, int G, int prec, std::vector<unsigned short>& out) {
  const int CT = convgn_ct(N, G);
  if (!CT || (cin & 63) || (prec != PREC_BF16 && prec != PREC_F16)) return hipErrorInvalidValue;
  const int NS = 32 * CT, K = 3 * cin, NKT = K / 64;
  out.clear();
  out.reserve((size_t)N * K);
  for (int sl = 0; sl < N / NS; ++sl)
    for (int kt = 0; kt < NKT; ++kt)
      for (int s = 0; s < 4; ++s)
        for (int ct = 0; ct < CT; ++ct)
          for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, hi = lane >> 5;
            const float* src = rows + (size_t)(sl * NS + 32 * ct + i) * K + kt * 64 + 16 * s + 8 * hi;
            for (int e = 0; e < 8; ++e) out.push_back(f32_to_op16_bits(src[e], prec));
          }
  return hipSuccess;
}

template <typename TM, int RT, int CT> static hipError_t launch_cg(const ConvGnArgs& a, hipStream_t s) {
  const int NS = 32 * CT;
  hipLaunchKernelGGL((convgn_kernel<TM, RT, CT>), dim3(a.B * (a.N / NS)), dim3(512), 0, s, a);
  return hipGetLastError();
}
template <typename TM> static hipError_t launch_cg_tm(const ConvGnArgs& a, int RT, int CT, hipStream_t s) {
#define NS2VC_CG(RT_, CT_) if (RT == RT_ && CT == CT_) return launch_cg<TM, RT_, CT_>(a, s)
  NS2VC_CG(1, 1); NS2VC_CG(2, 1); NS2VC_CG(3, 1); NS2VC_CG(4, 1);
  NS2VC_CG(1, 2); NS2VC_CG(2, 2); NS2VC_CG(3, 2);
  NS2VC_CG(1, 3); NS2VC_CG(2, 3);
#undef NS2VC_CG
  return hipErrorInvalidValue;
}

hipError_t launch_convgn(const ConvGnArgs& a, int prec, hipStream_t s) {
  if (!convgn_eligible(a.cin, a.N, a.G, a.T, prec) || a.B <= 0) return hipErrorInvalidValue;
  if (!a.a || !a.wpack || !a.bias || !a.gamma || !a.beta || !a.out_op) return hipErrorInvalidValue;
  if ((a.lda & 7) || a.lda < a.cin || (a.ldo & 7) || a.ldo < a.N) return hipErrorInvalidValue;
  if ((unsigned long long)a.B * a.T * a.lda * 2ull > 0xFFF00000ull) return hipErrorInvalidValue;
  if (a.temb && (a.ldtemb < a.temb_off + 2 * a.N)) return hipErrorInvalidValue;
  const int RT = (a.T + 255) / 256, CT = convgn_ct(a.N, a.G);
  return prec == PREC_BF16 ? launch_cg_tm<bf16_t>(a, RT, CT, s) : launch_cg_tm<f16_t>(a, RT, CT, s);
}

}  // namespace ns2vc
