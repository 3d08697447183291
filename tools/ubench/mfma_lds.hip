// Do two waves on one SIMD fill each other's LDS-wait gaps between dependent MFMA chains?
// loop body: 2 x ds_read_b128 -> 8 dependent v_mfma_f32_32x32x2_f32 using the loaded values.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: reads issued right before use (exposed latency), 1: reads prefetched one iteration ahead
__global__ void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 4];
  for (int i = threadIdx.x; i < 64 * 68 * 4; i += blockDim.x) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* base = lds + (wave & 3) * 64 * 68 + (lane & 31) * 68 + (lane >> 5) * 32;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  f32x4 a = *(const f32x4*)(base), b = *(const f32x4*)(base + 4);
  for (int i = 0; i < iters; ++i) {
    const int off = (i & 3) * 8;
    f32x4 na, nb;
    if (MODE == 1) { na = *(const f32x4*)(base + off); nb = *(const f32x4*)(base + off + 4); }
    else { a = *(const f32x4*)(base + off); b = *(const f32x4*)(base + off + 4); }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[c], a[c], acc, 0, 0, 0);
    if (MODE == 1) { a = na; b = nb; }
  }
  float s = 0;
  for (int e = 0; e < 16; ++e) s += acc[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int waves_per_simd) {
  float* d;
  hipMalloc(&d, 1 << 24);
  const int iters = 4000, threads = 256;
  const int blocks = 256 * waves_per_simd;  // co-resident BLOCKS of 4 waves (like the real kernels)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * 8.0 * iters * 2 * 32 * 32 * 2;
  printf("mode=%d blocks/CU=%d: %.1f TF/s (%.3f ms)\n", MODE, waves_per_simd, flops / ms / 1e9, ms);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) { run<0>(w); run<1>(w); }
  return 0;
}
