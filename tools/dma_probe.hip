// L2 -> LDS / VGPR streaming probe for gfx950 (MI355X): how many bytes per clock per CU does a GEMM operand
// loader get, as a function of the instruction (global_load_lds / buffer_load..lds / global_load_dwordx4),
// resident waves per CU, pieces in flight per wave, row stride of the 8 x 128-B piece and sharing between waves.
// Used to size the GEMM operand pipeline in ns2vc_amd/csrc/gemm.hip (results: profiles/dma_probe_r01.txt).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_probe tools/dma_probe.hip && tools/bin/dma_probe
//
// A "piece" is what one wave-instruction moves: 64 lanes x 16 B = 8 rows x 128 B (one GEMM tile pass slice).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void blds16(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

struct Args {
  const char* src;
  unsigned long long bytes;     // pool size
  unsigned row_stride;          // bytes between the 8 rows of a piece
  unsigned npool;               // pieces in the pool
  int pieces;                   // pieces per wave
  int shared;                   // 1: every wave walks the same piece sequence
  unsigned long long* out;      // [wave][2] start / end
};

// MODE 0 global_load_lds, 1 buffer_load lds, 2 global_load_dwordx4 -> VGPR
template <int MODE, int INF>
__global__ __launch_bounds__(256) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned gw = blockIdx.x * 4 + wave;
  const unsigned lds0 = (unsigned)(size_t)smem + wave * (INF * 1024);
  const unsigned lane_off = (lane >> 3) * a.row_stride + (lane & 7) * 16;
  const unsigned piece_bytes = 8 * a.row_stride;
  unsigned idx = a.shared ? 0u : (gw * 97u) % a.npool;
  i32x4_t rsrc;
  {
    const unsigned long long p = reinterpret_cast<unsigned long long>(a.src);
    rsrc.x = (int)(unsigned)p; rsrc.y = (int)(unsigned)(p >> 32); rsrc.z = (int)0xffffffffu; rsrc.w = 0x00020000;
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
  }
  u32x4_t sink = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int p = 0; p < a.pieces; ++p) {
    const unsigned off = idx * piece_bytes + lane_off;
    const unsigned dst = lds0 + (p % INF) * 1024;
    if constexpr (MODE == 0) {
      glds16(a.src + off, dst);
      wait_vmcnt<INF - 1>();
    } else if constexpr (MODE == 1) {
      blds16(rsrc, off, dst);
      wait_vmcnt<INF - 1>();
    } else {
      // INF independent loads in flight: issue INF, then consume (compiler-counted)
      u32x4_t v[INF];
#pragma unroll
      for (int q = 0; q < INF; ++q) {
        unsigned i2 = idx + 61u * q; i2 = i2 >= a.npool ? i2 - a.npool : i2;   // npool > 61*INF
        v[q] = *reinterpret_cast<const u32x4_t*>(a.src + i2 * piece_bytes + lane_off);
      }
#pragma unroll
      for (int q = 0; q < INF; ++q) sink ^= v[q];
      p += INF - 1;
      idx += 61u * (INF - 1); idx = idx >= a.npool ? idx - a.npool : idx;
    }
    idx += 61u; idx = idx >= a.npool ? idx - a.npool : idx;
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (MODE == 2 && sink.x == 0x12345u) a.out[0] = sink.y;   // keep the loads alive
  if (lane == 0) { a.out[2 * gw] = t0; a.out[2 * gw + 1] = t1; }
}

// Tile-structured stream = the GEMM K loop without its MFMAs: per K tile every wave issues P pieces into ring slot
// kt % S, waits until tile kt has landed (at most (S-2)*P younger pieces in flight), then the workgroup barriers.
template <int S, int P, int BAR>
__global__ __launch_bounds__(256) void probe_tiles(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned gw = blockIdx.x * 4 + wave;
  const unsigned lds0 = (unsigned)(size_t)smem + wave * (S * P * 1024);
  const unsigned lane_off = (lane >> 3) * a.row_stride + (lane & 7) * 16;
  const unsigned piece_bytes = 8 * a.row_stride;
  unsigned idx = (gw * 97u) % a.npool;
  const int nk = a.pieces / P;
  auto issue = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < P; ++q) {
      glds16(a.src + idx * piece_bytes + lane_off, lds0 + (slot * P + q) * 1024);
      idx += 61u; idx = idx >= a.npool ? idx - a.npool : idx;
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s);
  int slot = S - 1;
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<(S - 2) * P>();
    if (BAR) __builtin_amdgcn_s_barrier();
    issue(slot);
    if (++slot == S) slot = 0;
  }
  wait_vmcnt<0>();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) { a.out[2 * gw] = t0; a.out[2 * gw + 1] = t1; }
}

// semantics check: what does an out-of-range lane of `buffer_load_dwordx4 ... lds` leave in LDS?  (the GEMM relies on it
// writing zeros for padded / out-of-bounds rows); also checks the SGPR soffset form
__global__ void oob_check(const char* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned buf[256 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) buf[i] = 0xAAAAAAAAu;
  __syncthreads();
  i32x4_t rsrc;
  const unsigned long long p = reinterpret_cast<unsigned long long>(src);
  rsrc.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p); rsrc.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
  rsrc.z = (int)nbytes; rsrc.w = 0x00020000;
  const unsigned voff = (lane & 1) ? 0xFFFFFFF0u : (unsigned)lane * 16u;     // odd lanes out of range
  const unsigned lds = (unsigned)(size_t)buf;
  const unsigned soff = 1024;                                                // scalar offset: second KB of the source
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = buf[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int INF>
static void run(const char* name, char* pool, size_t pool_bytes, unsigned stride, int shared, int blocks_per_cu, unsigned long long* dout) {
  const int ncu = 256, pieces = 512;
  Args a;
  a.src = pool; a.bytes = pool_bytes; a.row_stride = stride;
  a.npool = (unsigned)(pool_bytes / (8 * (size_t)stride));
  a.npool = std::min(a.npool, 2048u);                                      // 2 MB of cache lines regardless of the stride
  a.pieces = pieces; a.shared = shared; a.out = dout;
  const int grid = ncu * blocks_per_cu;
  const size_t lds = std::max<size_t>(4 * INF * 1024, (size_t)(160 * 1024 / blocks_per_cu) - 1024);   // pin residency to blocks_per_cu
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, INF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE, INF>), dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * grid * 4);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int w = 0; w < grid * 4; ++w) sum += (double)(h[2 * w + 1] - h[2 * w]);
  const double cyc_piece = sum / (grid * 4) / pieces;
  const double bpc = blocks_per_cu * 4 * 1024.0 / cyc_piece;
  const double tbs = (double)grid * 4 * pieces * 1024.0 / (ms * 1e-3) / 1e12;
  printf("%-10s inflight %2d  waves/CU %2d  stride %5u  %s : %7.1f cyc/piece/wave  %6.1f B/clk/CU  %6.2f TB/s (event)\n", name, INF, blocks_per_cu * 4,
         stride, shared ? "shared " : "private", cyc_piece, bpc, tbs);
  fflush(stdout);
}

template <int S, int P, int BAR>
static void run_tiles(char* pool, size_t pool_bytes, unsigned stride, int blocks_per_cu, unsigned long long* dout, unsigned npool_max = 2048u) {
  const int ncu = 256, pieces = 96 * P;
  Args a;
  a.src = pool; a.bytes = pool_bytes; a.row_stride = stride;
  a.npool = std::min((unsigned)(pool_bytes / (8 * (size_t)stride)), npool_max);
  a.pieces = pieces; a.shared = 0; a.out = dout;
  const int grid = ncu * blocks_per_cu;
  const size_t lds = std::max<size_t>(4 * S * P * 1024, (size_t)(160 * 1024 / blocks_per_cu) - 1024);
  if (lds > 160 * 1024) return;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_tiles<S, P, BAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 2; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe_tiles<S, P, BAR>), dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * grid * 4);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int w = 0; w < grid * 4; ++w) sum += (double)(h[2 * w + 1] - h[2 * w]);
  const double cyc_tile = sum / (grid * 4) / (pieces / P);
  printf("tiles: pool %6.1f MB ring %d  pieces/wave/tile %d  barrier %d  waves/CU %2d  stride %5u : %7.1f cyc/tile  %6.1f B/clk/CU  %6.2f TB/s (event)\n",
         a.npool / 1024.0, S, P, BAR, blocks_per_cu * 4, stride, cyc_tile, blocks_per_cu * 4 * P * 1024.0 / cyc_tile, (double)grid * 4 * pieces * 1024.0 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

static void run_oob() {
  char* src; unsigned* out;
  CK(hipMalloc(&src, 4096)); CK(hipMalloc(&out, 1024));
  std::vector<unsigned> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = 0x1000u + i;
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(oob_check, dim3(1), dim3(64), 0, 0, src, 4096u, out);
  std::vector<unsigned> r(256);
  CK(hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost));
  int ok_valid = 0, zero_oob = 0, stale_oob = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      const unsigned v = r[l * 4 + e];
      if (l & 1) { zero_oob += v == 0; stale_oob += v == 0xAAAAAAAAu; }
      else ok_valid += v == 0x1000u + 256 + l * 4 + e;         // soffset 1024 B = 256 dwords
    }
  printf("buffer_load..lds out-of-range lanes: %d/128 dwords zero, %d/128 stale; in-range lanes with SGPR soffset: %d/128 correct\n", zero_oob, stale_oob, ok_valid);
  fflush(stdout);
}

// Does data read by one kernel stay in the XCD L2s for the next kernel on the same stream?  Launch A streams a 2 MB
// region once (each workgroup a different slice); launch B re-reads the same slices (same workgroup -> same XCD) or a
// region nobody touched.  Short kernels, so the cold-start cost is what is measured.
static void run_reuse(char* pool, size_t pool_bytes, unsigned long long* dout) {
  Args a;
  a.src = pool; a.bytes = pool_bytes; a.row_stride = 128; a.npool = 2048; a.pieces = 8; a.shared = 0; a.out = dout;
  Args b2 = a; b2.src = pool + (64u << 20);            // untouched region
  const size_t lds = 64 * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e[4];
  for (auto& x : e) CK(hipEventCreate(&x));
  for (int rep = 0; rep < 3; ++rep) {
    // evict: stream 512 MB through the caches
    Args ev = a; ev.src = pool + (128u << 20); ev.npool = 524288; ev.pieces = 256;
    hipLaunchKernelGGL((probe<0, 8>), dim3(512), dim3(256), lds, 0, ev);
    CK(hipEventRecord(e[0], 0));
    hipLaunchKernelGGL((probe<0, 8>), dim3(256), dim3(256), lds, 0, a);      // A: first touch
    CK(hipEventRecord(e[1], 0));
    hipLaunchKernelGGL((probe<0, 8>), dim3(256), dim3(256), lds, 0, a);      // B: same data, next kernel
    CK(hipEventRecord(e[2], 0));
    hipLaunchKernelGGL((probe<0, 8>), dim3(256), dim3(256), lds, 0, b2);     // C: cold data
    CK(hipEventRecord(e[3], 0));
    CK(hipEventSynchronize(e[3]));
    float t1, t2, t3;
    CK(hipEventElapsedTime(&t1, e[0], e[1])); CK(hipEventElapsedTime(&t2, e[1], e[2])); CK(hipEventElapsedTime(&t3, e[2], e[3]));
    printf("cross-kernel reuse (2 MB, 8 pieces/wave): first touch %.1f us, same data in the next kernel %.1f us, untouched data %.1f us\n", t1 * 1e3, t2 * 1e3, t3 * 1e3);
  }
  fflush(stdout);
}

int main(int argc, char** argv) {
  run_oob();
  const size_t pool_bytes = (size_t)786432 * 1024;   // 768 MB: 2048 pieces at the largest stride, or a pool beyond the 256 MB Infinity Cache
  char* pool;
  unsigned long long* dout;
  CK(hipMalloc(&pool, pool_bytes + 65536));
  CK(hipMemset(pool, 1, pool_bytes + 65536));
  CK(hipMalloc(&dout, 2 * 8 * 256 * 16 * 8));
  run_reuse(pool, pool_bytes, dout);
  // cold streams: pool larger than the 4 MB L2 of an XCD (-> Infinity Cache) and larger than the Infinity Cache (-> HBM)
  for (unsigned np : {2048u, 65536u, 786432u})
    for (int bpc : {1, 2, 3}) {
      run_tiles<2, 6, 1>(pool, pool_bytes, 128, bpc, dout, np);
      run_tiles<3, 6, 1>(pool, pool_bytes, 128, bpc, dout, np);
      run_tiles<4, 6, 1>(pool, pool_bytes, 128, bpc, dout, np);
      run_tiles<3, 4, 1>(pool, pool_bytes, 128, bpc, dout, np);
      run_tiles<4, 4, 1>(pool, pool_bytes, 128, bpc, dout, np);
      run_tiles<6, 4, 1>(pool, pool_bytes, 128, bpc, dout, np);
    }
  if (argc < 2) return 0;
  for (int bpc : {1, 2, 3}) {
    run_tiles<2, 6, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<2, 6, 0>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<3, 6, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<3, 6, 0>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<4, 6, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<2, 4, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<3, 4, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<4, 4, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<2, 8, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<3, 8, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<4, 8, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<6, 2, 1>(pool, pool_bytes, 768, bpc, dout);
    run_tiles<8, 2, 1>(pool, pool_bytes, 768, bpc, dout);
  }
  if (argc < 3) return 0;
  const unsigned strides[] = {128, 256, 768, 3072};
  for (int shared = 0; shared < 2; ++shared)
    for (unsigned st : strides)
      for (int bpc : {1, 2, 4}) {
        run<0, 4>("glds", pool, pool_bytes, st, shared, bpc, dout);
        run<0, 8>("glds", pool, pool_bytes, st, shared, bpc, dout);
        run<0, 16>("glds", pool, pool_bytes, st, shared, bpc, dout);
        run<1, 8>("buffer-lds", pool, pool_bytes, st, shared, bpc, dout);
        run<2, 4>("vgpr x4", pool, pool_bytes, st, shared, bpc, dout);
        run<2, 8>("vgpr x4", pool, pool_bytes, st, shared, bpc, dout);
      }
  return 0;
}
