// Do kernels on different HIP streams of one host thread overlap on MI355X?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = a;
}
int main() {
  float* d;
  hipMalloc(&d, 1024);
  const int NS = 4, K = 400;
  hipStream_t st[NS];
  for (int i = 0; i < NS; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  for (int blocks : {32, 128}) {
    for (int ns = 1; ns <= NS; ns *= 2) {
      for (int rep = 0; rep < 2; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k)
          for (int s = 0; s < ns; ++s) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, st[s], d, 20000);
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        if (rep)
          printf("blocks=%3d streams=%d: %d kernels/stream, enqueue %.2f ms, total %.2f ms (%.1f us per kernel-slot)\n", blocks, ns, K,
                 std::chrono::duration<double, std::milli>(t1 - t0).count(),
                 std::chrono::duration<double, std::milli>(t2 - t0).count(),
                 std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
      }
    }
  }
  return 0;
}
