// Per-kernel cost of a dependent chain of trivial launches on one stream: plain launches vs a captured
// hipGraph, for a few grid sizes and with / without a small global write.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

// write: 0 nothing, 1 plain store, 2 write-through (sc1) store, 3 plain 16-byte store, 4 write-through 16-byte store,
// 5 non-temporal store
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_empty(float* p, int write) {
  const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (write == 1) p[i] = 1.0f;
  else if (write == 2) __hip_atomic_store(p + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (write == 3) reinterpret_cast<f4*>(p)[i] = f4{1.f, 2.f, 3.f, 4.f};
  else if (write == 4) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<f4*>(p) + i), "v"(v) : "memory");
  } else if (write == 5) __builtin_nontemporal_store(1.0f, p + i);
}

static double run_stream(int grid, int write, float* d, hipStream_t st, int n) {
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d, write);
  hipStreamSynchronize(st);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d, write);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3 / n;
}

static double run_graph(int grid, int write, float* d, hipStream_t st, int n) {
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, d, write);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, st);
  hipGraphLaunch(ge, st);
  hipEventRecord(b, st);
  hipStreamSynchronize(st);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  return ms * 1e3 / n;
}

int main() {
  float* d;
  hipMalloc(&d, 1 << 26);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const int n = 1000;
  for (int write = 0; write < 6; ++write)
    for (int grid : {1, 256, 2048}) {
      printf("grid=%5d write=%d: stream %.2f us/kernel, graph %.2f us/kernel\n", grid, write,
             run_stream(grid, write, d, st, n), run_graph(grid, write, d, st, n));
    }
  return 0;
}
