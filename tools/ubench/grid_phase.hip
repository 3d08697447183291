// What does a grid-wide phase boundary cost INSIDE one persistent kernel on MI355X (256 CUs in 8 XCDs, one L2 per XCD)?
// The alternative to a kernel boundary (launch_floor.hip: ~5-8 us per dependent launch on the decode path's small steps).
//   barrier variants (G co-resident blocks of 256 threads, blockIdx % 8 = XCD):
//     A  one agent-scope atomic counter, everybody polls it
//     B  hierarchical: per-XCD counter (32 arrivals each), the last arriver of an XCD bumps a global counter (8 arrivals)
//     C  per-block epoch flags + a master block that gathers them and publishes a release word
//   phase variants: barrier + every block reads `rd_kb` KB written by OTHER blocks in the previous phase through sc1
//     (L2-bypassing) 16-byte loads, does `mfma` dependent 32x32x2 f32 MFMAs, writes 4 KB through sc1 stores.
// Every poll loop is bounded (no hang on a mistake): a timeout sets an error word that main() reports.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Sync {
  unsigned* ctr;       // [0] global counter, [16 + 16 * x] per-XCD counters (own cache lines)
  unsigned* flags;     // [G * 16] per-block epoch flags (own cache lines), [G * 16 + 16] release word
  unsigned* err;
};

#define SPIN_LIMIT (1u << 22)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void spin_until_ge(const unsigned* p, unsigned target, unsigned* err) {
  unsigned n = 0;
  while (ld_agent(p) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++n > SPIN_LIMIT) { atomicExch(err, 1u); break; }
  }
}

template <int VAR>
__device__ __forceinline__ void grid_barrier(const Sync& s, int G, unsigned epoch) {   // epoch = 1, 2, 3, ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (VAR == 0) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(s.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      spin_until_ge(s.ctr, epoch * (unsigned)G, s.err);
    }
  } else if (VAR == 1) {
    if (threadIdx.x == 0) {
      const int x = blockIdx.x & 7;
      const unsigned per = (unsigned)((G >> 3) + (x < (G & 7) ? 1 : 0));
      const unsigned old = __hip_atomic_fetch_add(s.ctr + 16 + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == epoch * per) __hip_atomic_fetch_add(s.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nx = (unsigned)(G < 8 ? G : 8);
      spin_until_ge(s.ctr, epoch * nx, s.err);
    }
  } else {
    if (threadIdx.x == 0) st_agent(s.flags + 16 * blockIdx.x, epoch);
    if (blockIdx.x == 0) {
      for (int b = threadIdx.x; b < G; b += blockDim.x) spin_until_ge(s.flags + 16 * b, epoch, s.err);
      __syncthreads();
      if (threadIdx.x == 0) st_agent(s.flags + 16 * G + 16, epoch);
    }
    if (threadIdx.x == 0) spin_until_ge(s.flags + 16 * G + 16, epoch, s.err);
  }
  __syncthreads();
}

template <int VAR>
__global__ __launch_bounds__(256) void barrier_only(Sync s, int G, int iters) {
  for (int i = 0; i < iters; ++i) grid_barrier<VAR>(s, G, (unsigned)(i + 1));
}

// phase: read rd_f4 float4 per thread from the region of block (b + 1 + i) % G (written in the previous phase), chain of
// MFMAs, write one float4 per thread into the own region
template <int VAR, int RD>
__global__ __launch_bounds__(256) void phases(Sync s, int G, int iters, float* buf, int n_mfma, float* sink) {
  const int tid = threadIdx.x;
  const size_t region = (size_t)256 * 4 * 16;   // floats per block region (16 float4 per thread)
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  float a = 1.0f + tid * 1e-3f, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
    const int src = (blockIdx.x + 1 + i) % G;
    const float* rp = buf + (size_t)src * region + tid * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    f32x4 t[RD];
#pragma unroll
    for (int q = 0; q < RD; ++q)   // all requests in flight together: ONE memory round trip
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[q]) : "v"(rp + (size_t)q * 1024) : "memory");
#pragma unroll
    for (int q = 0; q < RD; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[q])::"memory");   // the values are used AFTER the wait
#pragma unroll
    for (int q = 0; q < RD; ++q) v += t[q];
    a += v[0] * 1e-9f;
    for (int m = 0; m < n_mfma; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
    float* wp = buf + (size_t)blockIdx.x * region + tid * 4 + (size_t)(i & 15) * 1024;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp), "v"(o) : "memory");
    grid_barrier<VAR>(s, G, (unsigned)(i + 1));
  }
  if (acc[5] == 123.456f) sink[0] = acc[5];
}

// Correctness of the exchange under address RE-USE (the chain kernel's scratch buffers are rewritten every phase): in phase i
// every block overwrites the SAME 4 KB of its region with tag(i, block) through sc1 stores; after the boundary every block reads
// the regions of blocks b+1, b+9, b+17 (other XCDs and its own) through sc1 loads and counts words that are not tag(i, producer).
// A consumer L2 / L1 that served a stale line would show up here.  MODE 1: plain loads instead (expected to FAIL: shows the test bites).
template <int VAR, int PLAIN>
__global__ __launch_bounds__(256) void verify(Sync s, int G, int iters, float* buf, unsigned* bad) {
  const int tid = threadIdx.x;
  const size_t region = (size_t)256 * 4 * 16;
  unsigned nbad = 0;
  for (int i = 0; i < iters; ++i) {
    const float tag = (float)(i * 1024 + (int)blockIdx.x);
    f32x4 o = {tag, tag, tag, tag};
    float* wp = buf + (size_t)blockIdx.x * region + tid * 4;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp), "v"(o) : "memory");
    grid_barrier<VAR>(s, G, (unsigned)(2 * i + 1));
    for (int k = 0; k < 3; ++k) {
      const int src = (blockIdx.x + 1 + 8 * k) % G;
      const float want = (float)(i * 1024 + src);
      const float* rp = buf + (size_t)src * region + tid * 4;
      f32x4 t;
      if (PLAIN) {
        t = *reinterpret_cast<const volatile f32x4*>(rp);
      } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t) : "v"(rp) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t)::"memory");
      }
      nbad += (t[0] != want) + (t[1] != want) + (t[2] != want) + (t[3] != want);
    }
    grid_barrier<VAR>(s, G, (unsigned)(2 * i + 2));   // nobody overwrites before everybody has read
  }
  if (nbad) atomicAdd(bad, nbad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K, typename... A>
static double run(K kernel, int G, Sync s, hipStream_t st, A... args) {
  CK(hipMemsetAsync(s.ctr, 0, 4096, st));
  CK(hipMemsetAsync(s.flags, 0, (size_t)(G + 2) * 64, st));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, st));
  hipLaunchKernelGGL(kernel, dim3(G), dim3(256), 0, st, s, G, args...);
  CK(hipGetLastError());
  CK(hipEventRecord(b, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3;
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  Sync s;
  CK(hipMalloc(&s.ctr, 4096));
  CK(hipMalloc(&s.flags, 520 * 64));
  CK(hipMalloc(&s.err, 64));
  CK(hipMemset(s.err, 0, 64));
  float *buf, *sink;
  CK(hipMalloc(&buf, (size_t)512 * 256 * 4 * 16 * sizeof(float)));
  CK(hipMemset(buf, 0, (size_t)512 * 256 * 4 * 16 * sizeof(float)));
  CK(hipMalloc(&sink, 64));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int iters = 2000;
  for (int G : {8, 32, 64, 128, 256}) {
    const double t0 = run(barrier_only<0>, G, s, st, 1);            // launch + one barrier: the fixed part
    const double a = (run(barrier_only<0>, G, s, st, iters) - t0) / (iters - 1);
    const double b = (run(barrier_only<1>, G, s, st, iters) - t0) / (iters - 1);
    const double c = (run(barrier_only<2>, G, s, st, iters) - t0) / (iters - 1);
    printf("G=%3d  barrier only [us]: atomic counter %.2f | per-XCD + global %.2f | flags + master %.2f   (launch + 1 barrier %.1f us)\n",
           G, a, b, c, t0);
  }
  for (int G : {64, 256}) {
    for (int nm : {0, 64}) {
      printf("G=%3d  phase = 1 sc1 load round trip (1 / 4 / 16 x 16 B per thread), %2d MFMAs, 1 sc1 store, barrier [us]:\n", G, nm);
      printf("        counter   %.2f / %.2f / %.2f\n", run(phases<0, 1>, G, s, st, iters, buf, nm, sink) / iters,
             run(phases<0, 4>, G, s, st, iters, buf, nm, sink) / iters, run(phases<0, 16>, G, s, st, iters, buf, nm, sink) / iters);
      printf("        per-XCD   %.2f / %.2f / %.2f\n", run(phases<1, 1>, G, s, st, iters, buf, nm, sink) / iters,
             run(phases<1, 4>, G, s, st, iters, buf, nm, sink) / iters, run(phases<1, 16>, G, s, st, iters, buf, nm, sink) / iters);
      printf("        flags     %.2f / %.2f / %.2f\n", run(phases<2, 1>, G, s, st, iters, buf, nm, sink) / iters,
             run(phases<2, 4>, G, s, st, iters, buf, nm, sink) / iters, run(phases<2, 16>, G, s, st, iters, buf, nm, sink) / iters);
    }
  }
  {
    unsigned* bad;
    CK(hipMalloc(&bad, 64));
    for (int plain = 0; plain < 2; ++plain) {
      CK(hipMemset(bad, 0, 64));
      const double t = plain ? run(verify<2, 1>, 256, s, st, 2000, buf, bad) : run(verify<2, 0>, 256, s, st, 2000, buf, bad);
      unsigned nb = 0;
      CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      printf("verify (G=256, 2000 phases, same addresses rewritten every phase, 3 producers per consumer): %s loads: %u stale words of %u  (%.2f us per phase pair)\n",
             plain ? "PLAIN" : "sc1", nb, 2000u * 256u * 256u * 4u * 3u, t / 2000);
    }
  }
  unsigned err = 0;
  CK(hipMemcpy(&err, s.err, 4, hipMemcpyDeviceToHost));
  printf("spin timeouts: %u\n", err);
  return 0;
}
