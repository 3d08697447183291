// Micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 for 1/2/4 independent accumulator chains,
// 1..4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16 / CHAINS; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int e = 0; e < 16; ++e) s += acc[c][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int CHAINS>
void run(int waves_per_simd) {
  float* d;
  hipMalloc(&d, 1 << 24);
  int iters = 2000;
  int threads = 256 * waves_per_simd;  // 4 SIMDs x waves
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float cyc;
  hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  double mfma_per_wave = 16.0 * iters;
  double flops = 256.0 * 4 * waves_per_simd * mfma_per_wave * 2 * 32 * 32 * 2;
  printf("chains=%d waves/simd=%d : %.1f TF/s, %.1f clock64-ticks per MFMA per wave, %.3f ms\n", CHAINS, waves_per_simd,
         flops / ms / 1e9, cyc / mfma_per_wave, ms);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<1>(w);
    run<2>(w);
    run<4>(w);
  }
  return 0;
}
