// Sustained v_mfma_f32_32x32x16_bf16 rate: 4 accumulators per wave, 6 dependent MFMAs each per "slice"
// (the 3 x bf16 product pattern), with and without 12 ds_read_b128 + a block barrier per slice.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned g_rand = 0;   // 1: random bf16 operand bits (the chip clocks to its power budget: random operands toggle more)
template <int MODE>  // 0: MFMA only, 1: + 12 LDS reads per slice, 2: + barrier per slice
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 256 * 32];
  for (int i = threadIdx.x; i < 3 * 256 * 32 / 4; i += 256) {
    unsigned h = (i + 1) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // random mantissas and signs, exponents near 1.0 (finite products)
    ((unsigned*)lds)[i] = g_rand ? ((h & 0x807f807fu) | 0x3f003f00u) : 0x3c003c00u + i;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = lds + ((wave >> 1) * 64 + (lane & 31)) * 32 + (lane >> 5) * 16;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  u32x4 f[12];
  for (int q = 0; q < 12; ++q) f[q] = *(const u32x4*)(base + (q % 3) * 8192 + (q / 3) * 1024);
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
#pragma unroll
      for (int q = 0; q < 12; ++q) f[q] = *(const u32x4*)(base + (q % 3) * 8192 + ((q / 3 + it) & 3) * 1024);
    }
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[(t + a) % 6]),
                                                         __builtin_bit_cast(bf16x8, f[6 + (t * a) % 6]), acc[a], 0, 0, 0);
    if (MODE >= 2) __syncthreads();
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks_per_cu) {
  float* d;
  hipMalloc(&d, 1 << 22);
  const int iters = 2000, blocks = 256 * blocks_per_cu;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
  printf("mode %d, %d block(s)/CU: %.0f TF/s bf16 = %.0f TF/s fp32-equivalent (%.3f ms)\n", MODE, blocks_per_cu,
         flops / ms / 1e9, flops / ms / 1e9 / 6, ms);
  hipFree(d);
}

int main() {
  for (unsigned r = 0; r < 2; ++r) {
    hipMemcpyToSymbol(HIP_SYMBOL(g_rand), &r, sizeof(r));
    printf("operands: %s\n", r ? "random mantissas / signs" : "low-entropy constants");
    for (int b = 1; b <= 2; ++b) { run<0>(b); run<1>(b); run<2>(b); }
  }
  return 0;
}
