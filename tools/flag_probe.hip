// Price of a tile-local producer -> consumer dependency INSIDE one launch vs a kernel boundary, on MI355X (gfx950).
//
// the step graph has 210 kernel boundaries (~4 us each incl. grid fill / drain); the in-kernel wait tried in
// round 2 (GroupNorm producer) priced a whole-batch-item barrier (~1000 atomics on two cache lines), not a token-local
// hand-off.  This probe prices the token-local form, the one the chains attn1 -> to_out -> to_q -> attn2 -> feed-forward have:
//   consumer row tile t needs ONLY producer row tile t.
//
//   A  two launches per pair, stream order (what the engine does today), R pairs captured in one hipGraph
//   B  ONE launch per pair: blocks [0, nT) produce, blocks [nT, 2 nT) consume; consumer t first does its producer-independent
//      prologue (streams `wbytes` of "weights" into LDS, as a GEMM's first B tiles would), then waits on flag[t] (one lane,
//      relaxed agent-scope loads + s_sleep), one agent acquire, reads the tile
//   C  as B, producer publishes with write-through (sc1) stores + drained flag instead of plain stores + release fence
// Producers are dispatched first (lower block ids), and a launch is sized to be fully co-resident, so spinning is safe; every
// spin is bounded and a timeout is counted, not hung on.  Every word of the consumer's input is checked (stale data = error).
//
// build: hipcc --offload-arch=gfx950 -O3 -o flag_probe tools/flag_probe.hip ;  run: ./flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ROWS = 64, COLS = 128;                 // a 64 x 128 fp32 tile = 32 KB: what a level-0 GEMM workgroup writes
constexpr int TILE = ROWS * COLS;

struct Args {
  const float* in; float* mid; float* out; const float* w;
  unsigned* flag; unsigned* err; unsigned epoch;
  int nT, work, wbytes, mode;                        // mode 0: plain stores + release fence, 1: sc1 stores + drained flag
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16_sc1(float* p, float4 v4) {
  const f32x4_t v = {v4.x, v4.y, v4.z, v4.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float4 load16_sc1(const float* p) {
  f32x4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v.x, v.y, v.z, v.w);
}

// producer body: read the input tile, `work` dependent FMA rounds per element (stands for the K loop), write the tile
__device__ __forceinline__ void produce(const Args& a, int t, bool publish) {
  const int tid = threadIdx.x;
  const float* src = a.in + (size_t)t * TILE;
  float* dst = a.mid + (size_t)t * TILE;
  float4 v[TILE / 4 / 256];
#pragma unroll
  for (int i = 0; i < TILE / 4 / 256; ++i) v[i] = reinterpret_cast<const float4*>(src)[tid + 256 * i];
  for (int r = 0; r < a.work; ++r) {
#pragma unroll
    for (int i = 0; i < TILE / 4 / 256; ++i) { v[i].x = v[i].x * 1.0000001f + 1e-9f; v[i].y = v[i].y * 1.0000001f + 1e-9f; v[i].z = v[i].z * 1.0000001f + 1e-9f; v[i].w = v[i].w * 1.0000001f + 1e-9f; }
  }
  const float tag = (float)(a.epoch & 1023u);
#pragma unroll
  for (int i = 0; i < TILE / 4 / 256; ++i) {
    float4 o = make_float4(tag + (float)t, v[i].y, v[i].z, tag);      // .x / .w carry the epoch: a stale read is detectable
    if (publish && a.mode == 1) store16_sc1(dst + 4 * (tid + 256 * i), o);
    else reinterpret_cast<float4*>(dst)[tid + 256 * i] = o;
  }
  if (publish) {
    if (a.mode == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.flag + t, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(a.flag + t, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// consumer body: producer-independent prologue (stream wbytes into LDS), [wait], read the tile, check, write
__device__ __forceinline__ void consume(const Args& a, int t, bool wait) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int o = tid * 4; o < a.wbytes / 4; o += 256 * 4) {
    const float4 w = *reinterpret_cast<const float4*>(a.w + (size_t)((t * 4096 + o) & ((1 << 22) - 1)));
    *reinterpret_cast<float4*>(lds + (o & 8191 & ~3)) = w;
    acc += w.x;
  }
  if (wait) {
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(a.flag + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { atomicAdd(a.err, 1u << 16); break; }
      }
      if (a.mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
  const float* src = a.mid + (size_t)t * TILE;
  float* dst = a.out + (size_t)t * TILE;
  const float tag = (float)(a.epoch & 1023u);
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < TILE / 4 / 256; ++i) {
    const float4 v = (wait && a.mode == 1) ? load16_sc1(src + 4 * (tid + 256 * i)) : reinterpret_cast<const float4*>(src)[tid + 256 * i];
    bad += (v.x != tag + (float)t) || (v.w != tag);
    reinterpret_cast<float4*>(dst)[tid + 256 * i] = make_float4(v.x + acc * 0.f, v.y, v.z, v.w);
  }
  if (bad) atomicAdd(a.err, bad);
}

__global__ __launch_bounds__(256) void k_produce(const Args a) { produce(a, blockIdx.x, false); }
__global__ __launch_bounds__(256) void k_consume(const Args a) { consume(a, blockIdx.x, false); }
__global__ __launch_bounds__(256) void k_merged(const Args a) {
  if ((int)blockIdx.x < a.nT) produce(a, blockIdx.x, true);
  else consume(a, blockIdx.x - a.nT, true);
}

int main() {
  const int R = 40;                                   // pairs per captured graph
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *in, *mid, *out, *w;
  unsigned *flag, *err;
  const int maxT = 1024;
  CHECK(hipMalloc(&in, (size_t)maxT * TILE * 4)); CHECK(hipMalloc(&mid, (size_t)maxT * TILE * 4)); CHECK(hipMalloc(&out, (size_t)maxT * TILE * 4));
  CHECK(hipMalloc(&w, (size_t)(1 << 22) * 4 + (1 << 20))); CHECK(hipMalloc(&flag, maxT * 4)); CHECK(hipMalloc(&err, 4));
  CHECK(hipMemset(in, 0, (size_t)maxT * TILE * 4)); CHECK(hipMemset(w, 0, (size_t)(1 << 22) * 4 + (1 << 20)));
  CHECK(hipMemset(flag, 0, maxT * 4)); CHECK(hipMemset(err, 0, 4));
  printf("# tile-local producer -> consumer: two launches (A) vs one merged launch with per-tile flags (B plain+release, C sc1+drained flag)\n");
  printf("# %d pairs per captured graph, median of 9 graph launches; us per PAIR; errors = stale words + 65536 x timeouts\n", R);
  printf("%6s %6s %8s | %9s %9s %9s | %6s\n", "tiles", "work", "wbytes", "A 2-launch", "B flags", "C sc1", "errors");
  unsigned epoch = 1;
  for (int nT : {235, 469, 938}) {
    for (int work : {0, 200, 1000}) {
      for (int wbytes : {0, 32768}) {
        double us[3];
        unsigned total_err = 0;
        for (int variant = 0; variant < 3; ++variant) {
          hipGraph_t g; hipGraphExec_t ge;
          std::vector<double> t;
          for (int rep = 0; rep < 10; ++rep) {
            CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            for (int r = 0; r < R; ++r) {
              Args a{in, mid, out, w, flag, err, epoch++, nT, work, wbytes, variant == 2 ? 1 : 0};
              if (variant == 0) {
                hipLaunchKernelGGL(k_produce, dim3(nT), dim3(256), 0, s, a);
                hipLaunchKernelGGL(k_consume, dim3(nT), dim3(256), 0, s, a);
              } else {
                hipLaunchKernelGGL(k_merged, dim3(2 * nT), dim3(256), 0, s, a);
              }
            }
            CHECK(hipStreamEndCapture(s, &g));
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, s));
            CHECK(hipGraphLaunch(ge, s));
            CHECK(hipEventRecord(e1, s));
            CHECK(hipStreamSynchronize(s));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) t.push_back(ms * 1e3 / R);
            CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
            CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
          }
          std::sort(t.begin(), t.end());
          us[variant] = t[t.size() / 2];
          unsigned e = 0;
          CHECK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
          total_err += e;
          CHECK(hipMemset(err, 0, 4));
        }
        printf("%6d %6d %8d | %9.2f %9.2f %9.2f | %6u\n", nT, work, wbytes, us[0], us[1], us[2], total_err);
        fflush(stdout);
      }
    }
  }
  return 0;
}
