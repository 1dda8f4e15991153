// Repro for the r3 GroupNorm-prologue failure, second hypothesis (the first, tools/vmem_dma_race.hip, found nothing): the compiled
// prologue held    ds_read_b64 v[64:65], vA ; ds_read_b64 v[68:69], v52 ; ds_read_b64 v[70:71], vC ; s_waitcnt lgkmcnt(2) ;
//                  v_pk_mul_f32 v[52:53], v[30:31], v[64:65] op_sel:[0,1]
// i.e. a packed-fp32 VALU result written over the ADDRESS register (v52) of a DS instruction that has been issued but -- with the
// LDS pipe backed up by the sibling waves' LDS-DMA -- may not have read its address for the last 16 lanes yet.  In the failing
// launches v52 came out as ~0 in lanes 48-63 (the old address bits, a denormal float).  Does that sequence fail in isolation?
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ds_war_probe tools/ds_war_probe.hip && tools/bin/ds_war_probe
// MODE 0: the sequence above; 1: the same with a 32-bit v_mul_f32 writing v52; 2: packed result into other registers (control);
// 3: MODE 0 with s_waitcnt lgkmcnt(0) (the fix that was shipped).  DMA 1: waves 0-3 stream buffer_load_dwordx4 .. lds; 2: they also read fragments and issue MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4_t __attribute__((ext_vector_type(4)));
struct Args { const char* pool; unsigned pool_bytes; unsigned* out; int iters; };   // out: [0] wrong products, [1] wrong second reads, [2] records, [4..]

__device__ __forceinline__ void dma16(i32x4_t rsrc, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
template <int MODE, int DMA> __global__ __launch_bounds__(512) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float2* const tab = reinterpret_cast<float2*>(smem + 32768);                        // 64 (mean, rstd) pairs behind the DMA area
  if (tid < 64) tab[tid] = make_float2((float)tid, 0.5f + (float)tid / 64.f);
  __syncthreads();
  if (wave < 4) {
    if (!DMA) return;
    const unsigned long long p = reinterpret_cast<unsigned long long>(a.pool);
    i32x4_t rsrc = {__builtin_amdgcn_readfirstlane((int)(unsigned)p), __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32)), (int)a.pool_bytes, 0x00020000};
    unsigned off = ((blockIdx.x * 4 + wave) * 8192u) % a.pool_bytes;
    for (int it = 0; it < a.iters * 2; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dma16(rsrc, off + j * 1024 + lane * 16, (unsigned)(size_t)smem + wave * 8192 + j * 1024);
      if (DMA == 2) {                                                                  // ... and multiply like a GEMM consumer wave: fragment reads + MFMAs
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
        f32x16 acc = {};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const f16x8 fa = *reinterpret_cast<const f16x8*>(smem + wave * 8192 + ((lane * 16 + k * 1024) & 8191));
          const f16x8 fb = *reinterpret_cast<const f16x8*>(smem + ((wave * 8192 + 4096 + lane * 16 + k * 512) & 32767));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        }
        if (acc[0] == 12345.f) a.out[3] = 1;                                           // keep the chain alive
      }
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      off += 8192; if (off + 8192 > a.pool_bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  unsigned bad_p = 0, bad_r = 0;
  for (int it = 0; it < a.iters; ++it) {
    const int ia = (lane >> 2) + (it & 15), ib = (ia + 17) & 63, ic = (ia + 34) & 63;   // 4 lanes share an address, as in the prologue
    const unsigned aA = (unsigned)(size_t)(tab + (ia & 63)), aB = (unsigned)(size_t)(tab + ib), aC = (unsigned)(size_t)(tab + ic);
    float p0, p1, r0, r1;
    asm volatile("v_mov_b32 v30, 2.0\n\tv_mov_b32 v31, 4.0\n\tv_mov_b32 v52, %5\n\tv_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\t"
                 "ds_read_b64 v[64:65], %4\n\tds_read_b64 v[68:69], v52\n\tds_read_b64 v[70:71], %6\n\t"
                 ".if %7 == 3\n\ts_waitcnt lgkmcnt(0)\n\t.else\n\ts_waitcnt lgkmcnt(2)\n\t.endif\n\t"
                 ".if %7 == 1\n\tv_mul_f32 v52, v30, v65\n\tv_mul_f32 v53, v31, v65\n\t"
                 ".elseif %7 == 2\n\tv_pk_mul_f32 v[56:57], v[30:31], v[64:65] op_sel:[0,1]\n\tv_mov_b32 v52, v56\n\tv_mov_b32 v53, v57\n\t"
                 ".else\n\tv_pk_mul_f32 v[52:53], v[30:31], v[64:65] op_sel:[0,1]\n\t.endif\n\t"
                 "s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\tv_mov_b32 %0, v52\n\tv_mov_b32 %1, v53\n\tv_mov_b32 %2, v68\n\tv_mov_b32 %3, v69"
                 : "=&v"(p0), "=&v"(p1), "=&v"(r0), "=&v"(r1) : "v"(aA), "v"(aB), "v"(aC), "i"(MODE)
                 : "memory", "v30", "v31", "v52", "v53", "v56", "v57", "v64", "v65", "v68", "v69", "v70", "v71");
    const float rs = 0.5f + (float)(ia & 63) / 64.f;
    const bool bp = p0 != 2.0f * rs || p1 != 4.0f * rs, br = r0 != (float)ib || r1 != 0.5f + (float)ib / 64.f;
    bad_p += bp; bad_r += br;
    if (bp || br) {
      const unsigned slot = atomicAdd(a.out + 2, 1u);
      if (slot < 8) { unsigned* r = a.out + 4 + slot * 8; r[0] = blockIdx.x; r[1] = tid; r[2] = it; r[3] = __float_as_uint(p0); r[4] = __float_as_uint(p1);
        r[5] = __float_as_uint(2.0f * rs); r[6] = __float_as_uint(r0); r[7] = aB; }
    }
  }
  if (bad_p) atomicAdd(a.out + 0, bad_p);
  if (bad_r) atomicAdd(a.out + 1, bad_r);
}
template <int MODE, int DMA> static void run(const char* tag, Args a) {
  (void)hipMemset(a.out, 0, 4096);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
  for (int rep = 0; rep < 5; ++rep) probe<MODE, DMA><<<1024, 512, 40960>>>(a);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", tag); return; }
  std::vector<unsigned> h(1024); (void)hipMemcpy(h.data(), a.out, 4096, hipMemcpyDeviceToHost);
  printf("%-58s sequences %lld  wrong product %u  wrong 2nd read %u\n", tag, 5LL * 1024 * 256 * a.iters, h[0], h[1]);
  for (unsigned s = 0; s < h[2] && s < 4; ++s) { const unsigned* r = &h[4 + s * 8];
    printf("    block %u wave %u lane %u iter %u: v52 %08x v53 %08x want v52 %08x; 2nd read .x %08x (address register held %08x)\n", r[0], r[1] >> 6, r[1] & 63, r[2], r[3], r[4], r[5], r[6], r[7]); }
}
int main() {
  Args a; a.pool_bytes = 8u << 20; a.iters = 4000;
  (void)hipMalloc((void**)&a.pool, a.pool_bytes); (void)hipMemset((void*)a.pool, 0x11, a.pool_bytes); (void)hipMalloc((void**)&a.out, 4096);
  run<0, 0>("packed product over the DS address register, no DMA", a);
  run<0, 1>("packed product over the DS address register, DMA", a);
  run<1, 1>("32-bit products over the DS address register, DMA", a);
  run<2, 1>("packed product into other registers, DMA (control)", a);
  run<3, 1>("packed product over the address, lgkmcnt(0) first, DMA", a);
  run<0, 2>("packed product over the address, DMA + fragment reads + MFMA", a);
  run<2, 2>("packed product elsewhere, DMA + fragment reads + MFMA", a);
  run<1, 2>("32-bit products, DMA + fragment reads + MFMA", a);
  return 0;
}
