// r6: what bounds the K loops of the weight-streaming kernels -- L2 misses, or the L2 -> LDS path itself?
//
// Replays the address stream of conv3ts_kernel at level 3 (1024 -> 512 channels, M = 32 row blocks of 128 rows, 8 column groups of 64:
// 256 workgroups, workgroup id i on XCD i % 8 holding 4 row blocks x 8 column groups) WITHOUT MFMAs or fragment reads: per chunk step a
// workgroup pulls one 16 KB activation chunk (128 rows x 128 B at a 2 KB row stride; shared by the 8 column groups of its row block) and
// three 8 KB tile-major weight tiles (24 KB contiguous; shared by the 4 row blocks of the XCD) through LDS-DMA into a ring of S chunk
// slots, 8 loader waves, counted vmcnt + one barrier per chunk step -- the loop of csrc/convts.hip with its consumers removed.
//
// Variants:
//   rot R      the launch cycles over R copies of (weights, activations): R = 1 leaves the XCD's working set in its L2 between launches
//              (as far as 4 MB hold it), R = 12 (130 MB) makes every first touch of a launch come from the Infinity Cache
//   pf D       a ninth wave touches (one dword per 128-B line, never waited for) this workgroup's 1/32 share of the XCD's UNIQUE lines
//              of chunk step k + D: a run-ahead L2 prefetch, so that the loaders' requests D steps later are L2 hits
//   private    every workgroup streams its OWN tiles (no sharing inside the XCD, same bytes per workgroup): the r01 probe's pattern
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/convoy_probe tools/convoy_probe.hip && tools/bin/convoy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct Args {
  const char* w;            // [copy][chunk][col 8][24 KB]
  const char* a;            // [copy][row 4096][2 KB]
  unsigned long long w_copy, a_copy;   // bytes per copy
  int copy, nchunk, pf_dist, priv;
  unsigned long long* out;  // [wg][4]: t0, t1, xcc id, -
};

constexpr int CHUNK_W = 24 * 1024, CHUNK_A = 16 * 1024, PIECES = (CHUNK_W + CHUNK_A) / 1024;   // 40 pieces of 1 KB per chunk step
constexpr int NLOAD = 8;                                                                        // loader waves; 5 pieces each per chunk step
constexpr int PPW = PIECES / NLOAD;

template <int S, int PF>
__global__ __launch_bounds__(576) void convoy(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;          // 32 workgroups per XCD
  const int rb = a.priv ? (int)blockIdx.x : xcd * 4 + (idx >> 3); // activation row block (private: every workgroup its own)
  const int cg = a.priv ? 0 : (idx & 7);                           // weight column group
  const char* W = a.w + (size_t)a.copy * a.w_copy + (a.priv ? (size_t)blockIdx.x * a.nchunk * CHUNK_W : 0);
  const char* A = a.a + (size_t)a.copy * a.a_copy;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const size_t wstep = a.priv ? (size_t)CHUNK_W : (size_t)8 * CHUNK_W;

  auto issue = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int p = wave * PPW + q;                                 // piece 0..39 of the chunk step: 0..23 weights, 24..39 activations
      const char* src;
      if (p < 24) src = W + (size_t)c * wstep + (size_t)cg * CHUNK_W + p * 1024 + lane * 16;
      else src = A + ((size_t)rb * 128 + (p - 24) * 8 + (lane >> 3)) * 2048 + (size_t)c * 128 + (lane & 7) * 16;
      glds16(src, lds0 + (slot * PIECES + p) * 1024);
    }
  };
  unsigned sink = 0;
  auto touch = [&](int c) __attribute__((always_inline)) {
    // the XCD's unique lines of chunk step c: 8 col groups x 24 KB of weights (1536 lines) + 4 row blocks x 128 rows (512 lines);
    // this workgroup's share = 64 lines = one wave instruction
    const int j = idx * 64 + lane;
    const char* src;
    if (j < 1536) src = W + (size_t)c * wstep + (size_t)j * 128;
    else { const int r = j - 1536; src = A + ((size_t)(xcd * 4 + (r >> 7)) * 128 + (r & 127)) * 2048 + (size_t)c * 128; }
    // fixed destination register far above what the kernel allocates (checked: .vgpr_count is 121 only because of this clobber):
    // the load is never waited for, so the compiler must not be able to recycle its destination while it is in flight
    asm volatile("global_load_dword v120, %0, off" : : "v"(src) : "memory", "v120");
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < NLOAD) {
#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue(s, s);
    int slot = S - 1;
    for (int c = 0; c < a.nchunk; ++c) {
      wait_vmcnt<(S - 2) * PPW>();
      __builtin_amdgcn_s_barrier();
      if (c + S - 1 < a.nchunk) issue(c + S - 1, slot);
      else { /* keep the counted wait valid at the tail: issue nothing, wait for everything next time */ wait_vmcnt<0>(); }
      if (++slot == S) slot = 0;
    }
    wait_vmcnt<0>();
  } else {
    if (PF) for (int c = 0; c < a.pf_dist && c < a.nchunk; ++c) touch(c);
    for (int c = 0; c < a.nchunk; ++c) {
      __builtin_amdgcn_s_barrier();
      if (PF && c + a.pf_dist < a.nchunk) touch(c + a.pf_dist);
    }
    wait_vmcnt<0>();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.out[4 * blockIdx.x] = t0; a.out[4 * blockIdx.x + 1] = t1; a.out[4 * blockIdx.x + 2] = xcc & 15;
  }
  if (sink == 0x1234567u) a.out[3] = sink;
}

template <int S, int PF>
static void run(const char* w, const char* act, size_t w_copy, size_t a_copy, int rot, int nchunk, int pf_dist, int priv, unsigned long long* dout) {
  const int grid = 256;
  const size_t lds = (size_t)S * PIECES * 1024;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(convoy<S, PF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Args a;
  a.w = w; a.a = act; a.w_copy = w_copy; a.a_copy = a_copy; a.nchunk = nchunk; a.pf_dist = pf_dist; a.priv = priv; a.out = dout;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 48;
  for (int it = 0; it < 2 * rot; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy<S, PF>), dim3(grid), dim3(576), lds, 0, a); }
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < reps; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy<S, PF>), dim3(grid), dim3(576), lds, 0, a); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(4 * grid);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0; int same = 0;
  for (int g = 0; g < grid; ++g) { sum += (double)(h[4 * g + 1] - h[4 * g]); same += h[4 * g + 2] == h[4 * (g & 7) + 2]; }
  const double cyc = sum / grid / nchunk;     // s_memtime ticks at 100 MHz on gfx950: convert with the event time instead
  const double us = ms * 1e3 / reps;
  const double bytes = (double)grid * nchunk * PIECES * 1024.0;
  printf("%-8s ring %d  chunks %2d  rot %2d  pf %s%-2d : %6.2f us/launch  %6.1f ns/chunk-step  %5.1f B/clk/CU (launch)  %6.2f TB/s   [wg i on the XCD of wg i%%8: %d/256; %.0f ticks/step]\n",
         priv ? "private" : "shared", S, nchunk, rot, PF ? "" : "-", PF ? pf_dist : 0, us, us * 1e3 / nchunk, bytes / 256.0 / (us * 1e-6 * 2.4e9), bytes / (us * 1e-6) / 1e12, same, cyc);
  fflush(stdout);
}

// The same stream at conv3ts_kernel's OWN granularity: one step per tap (an 8 KB weight tile, issued DW steps ahead; the 16 KB activation chunk
// once per three steps, two chunks ahead), counted vmcnt + one s_barrier per step, 8 loader waves + NIDLE waves that only take part in the barriers
// (the consumers of the DMA-only ablation build, profiles/r05_ts_ablate.txt: 262 ns per step = 786 ns per chunk there).
template <int DW, int NIDLE>
__global__ __launch_bounds__(64 * (8 + NIDLE)) void convoy_tap(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int rb = xcd * 4 + (idx >> 3), cg = idx & 7;
  const char* W = a.w + (size_t)a.copy * a.w_copy;
  const char* A = a.a + (size_t)a.copy * a.a_copy;
  const unsigned lds0 = (unsigned)(size_t)smem;                    // [3 activation chunks][DW + 1 weight tiles]
  constexpr int SW = DW + 1;
  const int S = 3 * a.nchunk;
  auto issue_w = [&](int s, int slot) __attribute__((always_inline)) {      // tile of step s: chunk s / 3, tap s % 3 -> 8 KB, one piece per loader wave
    const int c = s / 3, tau = s - 3 * c;
    glds16(W + (size_t)c * 8 * CHUNK_W + (size_t)cg * CHUNK_W + tau * 8192 + wave * 1024 + lane * 16, lds0 + 3 * CHUNK_A + slot * 8192 + wave * 1024);
  };
  auto issue_a = [&](int c, int slot) __attribute__((always_inline)) {      // 16 KB chunk: two pieces per loader wave
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave * 2 + q;
      glds16(A + ((size_t)rb * 128 + p * 8 + (lane >> 3)) * 2048 + (size_t)c * 128 + (lane & 7) * 16, lds0 + slot * CHUNK_A + p * 1024);
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 8) {
#pragma unroll
    for (int d = 0; d < DW; ++d) issue_w(d, d);
    issue_a(0, 0); issue_a(1, 1);
    int wslot = DW % SW, aslot = 2;
    for (int s = 0; s < S; ++s) {
      const int tau = s % 3;
      // in flight behind tile s: tiles s+1 .. s+DW-1 (one piece each) and the chunks issued at steps s-DW+1 .. s-1 -- plus, at the head, the prologue's two chunks
      int allow = DW - 1;
      for (int b = 1; b < DW; ++b) allow += ((s - b) >= 0 && (s - b) % 3 == 0) ? 2 : 0;
      if (s < DW) allow = (tau == 0 && s / 3 < 2) ? 0 : allow;     // (head of the loop: be conservative while the prologue chunks are in flight)
      if (tau == 0) allow = allow;                                   // chunk s/3 was issued >= 6 steps ago when DW <= 6: older than tile s
      switch (allow) {
        case 0: wait_vmcnt<0>(); break; case 1: wait_vmcnt<1>(); break; case 2: wait_vmcnt<2>(); break; case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break; case 5: wait_vmcnt<5>(); break; case 6: wait_vmcnt<6>(); break; case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break; case 9: wait_vmcnt<9>(); break; default: wait_vmcnt<10>(); break;
      }
      __builtin_amdgcn_s_barrier();
      if (s + DW < S) issue_w(s + DW, wslot);
      if (++wslot == SW) wslot = 0;
      if (tau == 0 && s / 3 + 2 < a.nchunk && s > 0) { issue_a(s / 3 + 2, aslot); if (++aslot == 3) aslot = 0; }
      else if (tau == 0 && s == 0) { /* chunks 0 and 1 came with the prologue; chunk 2 goes out at step 0 into slot 2 */ if (2 < a.nchunk) issue_a(2, 2); aslot = 0; }
    }
    wait_vmcnt<0>();
  } else {
    for (int s = 0; s < S; ++s) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    a.out[4 * blockIdx.x] = t0; a.out[4 * blockIdx.x + 1] = t1; a.out[4 * blockIdx.x + 2] = xcc & 15;
  }
}

template <int DW, int NIDLE>
static void run_tap(const char* w, const char* act, size_t w_copy, size_t a_copy, int rot, int nchunk, unsigned long long* dout) {
  const int grid = 256;
  const size_t lds = (size_t)3 * CHUNK_A + (size_t)(DW + 1) * 8192;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(convoy_tap<DW, NIDLE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Args a;
  a.w = w; a.a = act; a.w_copy = w_copy; a.a_copy = a_copy; a.nchunk = nchunk; a.pf_dist = 0; a.priv = 0; a.out = dout;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 48;
  for (int it = 0; it < 2 * rot; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy_tap<DW, NIDLE>), dim3(grid), dim3(64 * (8 + NIDLE)), lds, 0, a); }
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < reps; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy_tap<DW, NIDLE>), dim3(grid), dim3(64 * (8 + NIDLE)), lds, 0, a); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(4 * grid);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int g = 0; g < grid; ++g) sum += (double)(h[4 * g + 1] - h[4 * g]);
  printf("  (in-loop %.0f cyc/chunk = %.1f B/clk/CU) ", sum / grid / nchunk, 40960.0 / (sum / grid / nchunk));
  const double us = ms * 1e3 / reps;
  const double bytes = (double)grid * nchunk * PIECES * 1024.0;
  printf("tap-step ring: weights %d ahead, %d idle waves  chunks %2d  rot %2d : %6.2f us/launch  %6.1f ns/chunk (%5.1f ns/step)  %5.1f B/clk/CU (launch)  %6.2f TB/s\n",
         DW, NIDLE, nchunk, rot, us, us * 1e3 / nchunk, us * 1e3 / nchunk / 3, bytes / 256.0 / (us * 1e-6 * 2.4e9), bytes / (us * 1e-6) / 1e12);
  fflush(stdout);
}

// Chunk-granular steps as a conv3ts_kernel could run them: ONE barrier per 64-channel chunk; per chunk the 16 KB activation chunk (ring of 3: issued two
// chunks ahead) and the chunk's three weight tiles (BNT x 24 KB; ring of SWC chunk slots: issued SWC - 1 chunks ahead); four more waves stand in for the
// consumers: after the barrier they are busy for ~SLEEP x 64 cycles (s_sleep) before they reach the next barrier -- the slot they "read" cannot be refilled earlier.
template <int BNT, int SWC, int SLEEP, int SWZ = 0>
__global__ __launch_bounds__(768) void convoy_chunk(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  constexpr int NCG = 8 / BNT;                                      // column groups of 64 * BNT; row blocks per XCD = 32 / NCG
  const int rb = xcd * (32 / NCG) + idx / NCG, cg = idx % NCG;
  const char* W = a.w + (size_t)a.copy * a.w_copy;
  const char* A = a.a + (size_t)a.copy * a.a_copy;
  const unsigned lds0 = (unsigned)(size_t)smem;                    // [3 activation chunks][SWC weight chunk slots]
  constexpr int WB = BNT * CHUNK_W, WP = WB / 1024 / 8;             // weight bytes per chunk; pieces per loader wave
  auto issue_w = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < WP; ++q)
      glds16(W + (size_t)c * 8 * CHUNK_W + (size_t)cg * WB + (wave * WP + q) * 1024 + lane * 16, lds0 + 3 * CHUNK_A + slot * WB + (wave * WP + q) * 1024);
  };
  auto issue_a = [&](int c, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int p = wave * 2 + q;
      const int row = p * 8 + (lane >> 3);
      const int chk = SWZ ? ((lane & 7) ^ ((row >> 1) & 7)) : (lane & 7);      // SWZ: conv3ts_kernel's source-side XOR swizzle of the 16-B pieces inside a row's 128-B line
      glds16(A + ((size_t)rb * 128 + row) * 2048 + (size_t)c * 128 + chk * 16, lds0 + slot * CHUNK_A + p * 1024);
    }
  };
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (wave < 8) {
    // prologue: weights of chunks 0 .. SWC-2, then rows of chunks 0 and 1
#pragma unroll
    for (int d = 0; d < SWC - 1; ++d) issue_w(d, d);
    issue_a(0, 0); issue_a(1, 1);
    for (int c = 0; c < a.nchunk; ++c) {
      // needed: W(c), A(c).  Issue order per step: W(c + SWC - 1) then A(c + 2).
      if (c == 0) { wait_vmcnt<2>(); }                               // all but A(1)
      else if (SWC == 2) wait_vmcnt<2>();                            // behind W(c) [step c-1]: A(c+1)
      else wait_vmcnt<WP + 2>();                                     // SWC == 3: behind W(c), A(c) [step c-2]: W(c+1), A(c+1)
      __builtin_amdgcn_s_barrier();
      if (c + SWC - 1 < a.nchunk) issue_w(c + SWC - 1, (c + SWC - 1) % SWC);
      if (c + 2 < a.nchunk) issue_a(c + 2, (c + 2) % 3);
      if (c + SWC - 1 >= a.nchunk || c + 2 >= a.nchunk) wait_vmcnt<0>();   // tail: keep the counted waits valid
    }
    wait_vmcnt<0>();
  } else {
    for (int c = 0; c < a.nchunk; ++c) {
      __builtin_amdgcn_s_barrier();
      if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) { a.out[4 * blockIdx.x] = t0; a.out[4 * blockIdx.x + 1] = t1; }
}

template <int BNT, int SWC, int SLEEP, int SWZ = 0>
static void run_chunk(const char* w, const char* act, size_t w_copy, size_t a_copy, int rot, int nchunk, unsigned long long* dout) {
  const int grid = 256;
  const size_t lds = (size_t)3 * CHUNK_A + (size_t)SWC * BNT * CHUNK_W;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(convoy_chunk<BNT, SWC, SLEEP, SWZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  Args a;
  a.w = w; a.a = act; a.w_copy = w_copy; a.a_copy = a_copy; a.nchunk = nchunk; a.pf_dist = 0; a.priv = 0; a.out = dout;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 48;
  for (int it = 0; it < 2 * rot; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy_chunk<BNT, SWC, SLEEP, SWZ>), dim3(grid), dim3(768), lds, 0, a); }
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < reps; ++it) { a.copy = it % rot; hipLaunchKernelGGL((convoy_chunk<BNT, SWC, SLEEP, SWZ>), dim3(grid), dim3(768), lds, 0, a); }
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(4 * grid);
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int g = 0; g < grid; ++g) sum += (double)(h[4 * g + 1] - h[4 * g]);
  const double us = ms * 1e3 / reps;
  const double bpc = 16 * 1024.0 + BNT * CHUNK_W;
  if (SWZ) printf("[rows XOR-swizzled at the source] ");
  printf("chunk-step: BN %3d  weight slots %d (LDS %3zu KB)  consumers busy ~%4d cyc  chunks %2d  rot %2d : %6.2f us/launch  %6.1f ns/chunk  in-loop %6.0f cyc/chunk = %5.1f B/clk/CU\n",
         64 * BNT, SWC, lds / 1024, SLEEP * 64, nchunk, rot, us, us * 1e3 / nchunk, sum / grid / nchunk, bpc / (sum / grid / nchunk));
  fflush(stdout);
}

int main() {
  const int NCH = 16, ROT = 12;
  const size_t w_copy = (size_t)256 * NCH * CHUNK_W;            // private mode needs 256 x the shared weights; shared uses the first 8 x NCH tiles
  const size_t a_copy = (size_t)256 * 128 * 2048;               // 256 row blocks (private) / 32 (shared) x 128 rows x 2 KB
  char *w, *act;
  unsigned long long* dout;
  CK(hipMalloc(&w, w_copy * ROT)); CK(hipMalloc(&act, a_copy * ROT));
  CK(hipMemset(w, 1, w_copy * ROT)); CK(hipMemset(act, 1, a_copy * ROT));
  CK(hipMalloc(&dout, 4 * 256 * 8));
  printf("# conv3ts level-3 stream (1024 -> 512, 16 chunk steps of 40 KB per workgroup, 256 workgroups, 8 loader waves), DMA only\n");
  for (int rot : {1, ROT}) {
    run<2, 0>(w, act, w_copy, a_copy, rot, NCH, 0, 0, dout);
    run<3, 0>(w, act, w_copy, a_copy, rot, NCH, 0, 0, dout);
    for (int d : {1, 2, 3, 4, 6}) run<3, 1>(w, act, w_copy, a_copy, rot, NCH, d, 0, dout);
    run<2, 1>(w, act, w_copy, a_copy, rot, NCH, 3, 0, dout);
    run<3, 0>(w, act, w_copy, a_copy, rot, NCH, 0, 1, dout);
  }
  for (int rot : {1, ROT}) {
    run_tap<2, 0>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_tap<2, 4>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_tap<3, 4>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_tap<5, 4>(w, act, w_copy, a_copy, rot, NCH, dout);
    run<3, 0>(w, act, w_copy, a_copy, rot, NCH, 0, 0, dout);
  }
  for (int rot : {1, ROT}) {
    run_chunk<1, 3, 0>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 3, 0, 1>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 2, 0>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 3, 12>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 2, 12>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 3, 19>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<1, 2, 19>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<2, 2, 0>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<2, 2, 19>(w, act, w_copy, a_copy, rot, NCH, dout);
    run_chunk<2, 2, 30>(w, act, w_copy, a_copy, rot, NCH, dout);
  }
  // half the K: the XCD's working set (1.5 MB of weights + 0.5 MB of rows) certainly fits its L2
  for (int rot : {1, ROT}) {
    run<3, 0>(w, act, w_copy, a_copy, rot, 8, 0, 0, dout);
    run<3, 1>(w, act, w_copy, a_copy, rot, 8, 3, 0, dout);
  }
  return 0;
}
