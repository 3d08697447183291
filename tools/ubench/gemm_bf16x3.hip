// Feasibility probe: fp32-accurate GEMM on the bf16 matrix cores.  Every fp32 operand is split exactly
// into three bf16 terms (x = x1 + x2 + x3 up to 2^-25 |x|); C = sum of the six products whose weight is
// >= 2^-16 (x1y1, x1y2, x2y1, x1y3, x2y2, x3y1), accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// Peak: 2.5 PF/s dense bf16 / 6 = 417 TF/s fp32-equivalent vs 157 TF/s for v_mfma_f32_32x32x2_f32.
//   ./gemm_bf16x3            -> accuracy vs fp64 + timing at the decode path's shapes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void split_kernel(const float* x, __bf16* p, size_t n) {  // p: [3][n]
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  __bf16 b1 = (__bf16)v;
  float r = v - (float)b1;
  __bf16 b2 = (__bf16)r;
  float r2 = r - (float)b2;
  __bf16 b3 = (__bf16)r2;
  p[i] = b1; p[n + i] = b2; p[2 * n + i] = b3;
}

// C[M,N] = A[M,K] * W[N,K]^T from split planes.  128x128 block tile, 4 waves x (64x64), BK = 16.
constexpr int BM = 128, BN = 128, BK = 16, ROWB = 48;  // LDS row: 16 bf16 + 16 B pad
constexpr int PLANE_B = (BM + BN) * ROWB, BUF_B = 3 * PLANE_B;

template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const __bf16* Ap, const __bf16* Wp, float* C, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
  const int tiles_n = N / BN;
  const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const size_t planeA = (size_t)M * K, planeW = (size_t)N * K;

  // staging: 6 x 16-byte loads per thread per slice: idx = tid + 256*q -> (plane, row, half-row)
  const __bf16* src[6];
  int dst[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int idx = tid + 256 * q, plane = idx / 512, rem = idx % 512, row = rem >> 1, hr = rem & 1;
    src[q] = (row < BM) ? Ap + plane * planeA + (size_t)(m0 + row) * K + hr * 8
                        : Wp + plane * planeW + (size_t)(n0 + row - BM) * K + hr * 8;
    dst[q] = plane * PLANE_B + row * ROWB + hr * 16;
  }
  f32x4 st[6];
  auto gload = [&](int s) {
#pragma unroll
    for (int q = 0; q < 6; ++q) st[q] = *reinterpret_cast<const f32x4*>(src[q] + s * BK);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<f32x4*>(lds + buf * BUF_B + dst[q]) = st[q];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nsl = K / BK;
  gload(0);
  lstore(0);
  gload(1);
  __syncthreads();
  for (int s = 0; s < nsl; ++s) {
    const unsigned char* base = lds + (s & 1) * BUF_B;
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = *reinterpret_cast<const bf16x8*>(base + p * PLANE_B + (wm0 + i * 32 + l32) * ROWB + half * 16);
        b[p][i] = *reinterpret_cast<const bf16x8*>(base + p * PLANE_B + (BM + wn0 + i * 32 + l32) * ROWB + half * 16);
      }
    if (s + 1 < nsl) lstore((s + 1) & 1);
    if (s + 2 < nsl) gload(s + 2);
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};  // small terms first
#pragma unroll
    for (int t = 6 - NPROD; t < 6; ++t)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]][mi], b[PB[t]][ni], acc[mi][ni], 0, 0, 0);
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
        const int col = n0 + wn0 + ni * 32 + l32;
        C[(size_t)row * N + col] = acc[mi][ni][e];
      }
}

template <int NPROD>
float run(const __bf16* Ap, const __bf16* Wp, float* C, int M, int N, int K, int iters) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<NPROD>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_B));
  const int grid = (M / BM) * (N / BN);
  hipLaunchKernelGGL(gemm_kernel<NPROD>, dim3(grid), dim3(256), 2 * BUF_B, 0, Ap, Wp, C, M, N, K);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_kernel<NPROD>, dim3(grid), dim3(256), 2 * BUF_B, 0, Ap, Wp, C, M, N, K);
  hipEventRecord(e1);
  CHECK(hipDeviceSynchronize());
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  const int Ms[] = {1024, 4096, 9216, 16384}, Ns[] = {512, 1536};
  const int K = 512;
  for (int M : Ms)
    for (int N : Ns) {
      std::vector<float> ha((size_t)M * K), hw((size_t)N * K);
      srand(1);
      for (auto& v : ha) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
      for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
      float *dA, *dW, *dC;
      __bf16 *pA, *pW;
      CHECK(hipMalloc(&dA, ha.size() * 4)); CHECK(hipMalloc(&dW, hw.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
      CHECK(hipMalloc(&pA, ha.size() * 6)); CHECK(hipMalloc(&pW, hw.size() * 6));
      CHECK(hipMemcpy(dA, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(dW, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(split_kernel, dim3((ha.size() + 255) / 256), dim3(256), 0, 0, dA, pA, ha.size());
      hipLaunchKernelGGL(split_kernel, dim3((hw.size() + 255) / 256), dim3(256), 0, 0, dW, pW, hw.size());
      const float ms6 = run<6>(pA, pW, dC, M, N, K, 20);
      std::vector<float> hc((size_t)M * N);
      CHECK(hipMemcpy(hc.data(), dC, hc.size() * 4, hipMemcpyDeviceToHost));
      // accuracy on a sample of entries vs fp64, next to a plain fp32 dot product
      double e6 = 0, e32 = 0, ref_mag = 0;
      for (int t = 0; t < 2000; ++t) {
        const int i = rand() % M, j = rand() % N;
        double r = 0; float f = 0;
        for (int k = 0; k < K; ++k) { r += (double)ha[(size_t)i * K + k] * hw[(size_t)j * K + k]; f += ha[(size_t)i * K + k] * hw[(size_t)j * K + k]; }
        e6 = fmax(e6, fabs(hc[(size_t)i * N + j] - r)); e32 = fmax(e32, fabs(f - r)); ref_mag = fmax(ref_mag, fabs(r));
      }
      const float ms3 = run<3>(pA, pW, dC, M, N, K, 20);
      const double fl = 2.0 * M * N * K;
      printf("M=%5d N=%4d K=%d: 6 products %.1f us (%.1f TF/s fp32-equiv), 3 products %.1f us (%.1f TF/s); max abs err 6p %.2e, fp32 loop %.2e (|C| <= %.1f)\n",
             M, N, K, ms6 * 1e3, fl / ms6 / 1e9, ms3 * 1e3, fl / ms3 / 1e9, e6, e32, ref_mag);
      hipFree(dA); hipFree(dW); hipFree(dC); hipFree(pA); hipFree(pW);
    }
  return 0;
}
