// Per-kernel cost of a dependent chain of trivial launches on one stream: plain launches vs a captured
// hipGraph, for a few grid sizes and with / without a small global write.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>

__global__ void k_empty(float* p, int write) {
  if (write) p[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f;
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
  for (int write = 0; write < 2; ++write)
    for (int grid : {1, 256, 2048}) {
      printf("grid=%5d write=%d: stream %.2f us/kernel, graph %.2f us/kernel\n", grid, write,
             run_stream(grid, write, d, st, n), run_graph(grid, write, d, st, n));
    }
  return 0;
}
