// Repro for the "gamma reads 0.0 in one 16-lane pass" failure of the r3 GroupNorm-prologue experiment
// (profiles/r03_gn_prologue_experiment.txt): does a global_load_dwordx4 -> VGPR of a wave that issues NO LDS-DMA itself
// return stale register content right after its s_waitcnt when SIBLING waves of the CU stream buffer_load..lds DMA?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/vmem_dma_race tools/vmem_dma_race.hip && tools/bin/vmem_dma_race
// Waves 0-3 of every workgroup stream LDS-DMA (MODE 1: dwordx4, 2: dword, 0: idle).  Waves 4-7 loop: clear 4 VGPRs, load a
// known 16 B per lane into them, s_waitcnt vmcnt(0), [NOPS x s_nop 7], snapshot (e0), s_sleep, snapshot again (e1).
// e0 wrong + e1 right = the write pass landed AFTER vmcnt said so (race); e0 and e1 wrong = the pass was lost / zero.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct Args { const char* pool; unsigned pool_bytes; const u32x4_t* table; unsigned* out; int iters; };   // out: [0] early bad, [1] late bad, [2..] records

template <int W> __device__ __forceinline__ void dma(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if constexpr (W == 4)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
template <int MODE, int NOPS> __global__ __launch_bounds__(512) void race(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {                                   // loaders: 8 pieces in flight, like the GEMM ring
    if (MODE == 0) return;
    const unsigned long long p = reinterpret_cast<unsigned long long>(a.pool);
    i32x4_t rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)p), __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)), (int)a.pool_bytes, 0x00020000};
    const unsigned lds0 = (unsigned)(size_t)smem + wave * 8192;
    unsigned off = ((blockIdx.x * 4 + wave) * 8192u) % a.pool_bytes;
    for (int it = 0; it < a.iters * 4; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma<MODE == 1 ? 4 : 1>(rsrc, off + j * 1024 + lane * (MODE == 1 ? 16 : 4), lds0 + j * 1024);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      off += 8192; if (off + 8192 > a.pool_bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const u32x4_t* src = a.table + (lane & 31);
  const u32x4_t want = {0x3f800000u + 4 * (lane & 31), 0x3f800001u + 4 * (lane & 31), 0x3f800002u + 4 * (lane & 31), 0x3f800003u + 4 * (lane & 31)};
  unsigned early = 0, late = 0;
  for (int it = 0; it < a.iters; ++it) {
    u32x4_t e0, e1;                                 // (one asm block: nothing of the compiler's between the wait and the first read)
    asm volatile("v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
                 "global_load_dwordx4 v[100:103], %8, off\n\ts_waitcnt vmcnt(0)\n\t.rept %9\n\ts_nop 7\n\t.endr\n\t"
                 "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\ts_sleep 4\n\t"
                 "v_mov_b32 %4, v100\n\tv_mov_b32 %5, v101\n\tv_mov_b32 %6, v102\n\tv_mov_b32 %7, v103"
                 : "=&v"(e0.x), "=&v"(e0.y), "=&v"(e0.z), "=&v"(e0.w), "=&v"(e1.x), "=&v"(e1.y), "=&v"(e1.z), "=&v"(e1.w)
                 : "v"(src), "i"(NOPS) : "memory", "v100", "v101", "v102", "v103");
    const bool b0 = e0.x != want.x || e0.y != want.y || e0.z != want.z || e0.w != want.w;
    const bool b1 = e1.x != want.x || e1.y != want.y || e1.z != want.z || e1.w != want.w;
    early += b0; late += b1;
    if (b0 || b1) {
      const unsigned slot = atomicAdd(a.out + 2, 1u);
      if (slot < 16) { unsigned* r = a.out + 4 + slot * 12; r[0] = blockIdx.x; r[1] = tid; r[2] = it; r[3] = b1;
        r[4] = e0.x; r[5] = e0.y; r[6] = e0.z; r[7] = e0.w; r[8] = e1.x; r[9] = e1.y; r[10] = e1.z; r[11] = e1.w; }
    }
  }
  if (early) atomicAdd(a.out + 0, early);
  if (late) atomicAdd(a.out + 1, late);
}
template <int MODE, int NOPS> static void run(const char* tag, Args a, int blocks) {
  hipMemset(a.out, 0, 4096);
  hipFuncSetAttribute(reinterpret_cast<const void*>(race<MODE, NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  for (int rep = 0; rep < 5; ++rep) race<MODE, NOPS><<<blocks, 512, 32768>>>(a);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", tag); return; }
  std::vector<unsigned> h(1024); hipMemcpy(h.data(), a.out, 4096, hipMemcpyDeviceToHost);
  printf("%-34s loads %lld  wrong right after the wait %u  still wrong after s_sleep %u\n", tag, 5LL * blocks * 256 * a.iters, h[0], h[1]);
  for (unsigned s = 0; s < h[2] && s < 6; ++s) { const unsigned* r = &h[4 + s * 12];
    printf("    block %u tid %u (wave %u lane %u) iter %u: e0 %08x %08x %08x %08x  e1 %08x %08x %08x %08x\n", r[0], r[1], r[1] >> 6, r[1] & 63, r[2], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]); }
}
int main() {
  Args a; a.pool_bytes = 8u << 20; a.iters = 2000;
  hipMalloc((void**)&a.pool, a.pool_bytes); hipMemset((void*)a.pool, 0x11, a.pool_bytes);
  std::vector<unsigned> tab(128); for (int i = 0; i < 128; ++i) tab[i] = 0x3f800000u + i;
  hipMalloc((void**)&a.table, 512); hipMemcpy((void*)a.table, tab.data(), 512, hipMemcpyHostToDevice);
  hipMalloc((void**)&a.out, 4096);
  const int blocks = 1024;
  run<0, 0>("no DMA (control)", a, blocks);
  run<1, 0>("sibling waves DMA dwordx4", a, blocks);
  run<2, 0>("sibling waves DMA dword", a, blocks);
  run<1, 1>("dwordx4 DMA, s_nop 7 before use", a, blocks);
  run<1, 4>("dwordx4 DMA, 4 x s_nop 7 before use", a, blocks);
  return 0;
}
