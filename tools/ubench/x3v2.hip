// Round-4 design probe for the 3 x bf16 split projection kernel ("x3 v2"): main-loop rate of a form in which
//   * the weight planes AND the activations reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers,
//     no ds_write, three slices in flight),
//   * the activations are either fp32 rows that every wave splits into bf16 terms AFTER reading its own 32 rows
//     back from LDS (APL = false; a wave owns its rows: 4 x 1 wave grid, wave tile 32 x 128), or pre-split planes
//     (APL = true: what a producer epilogue would have to write),
//   * the accumulators are transposed (D[n][m] = W frag x A frag: a lane holds 4 consecutive output columns of ONE
//     row, so the epilogue moves 16-byte pieces),
//   * W / A fragments of slice s+1 are read while the MFMAs of slice s issue (second fragment set in registers).
// Whole tiles per block (no stream-K): judge the rate on shapes with many tiles per block.
//   hipcc -O3 --offload-arch=gfx950 -o build_ub/x3v2 tools/ubench/x3v2.hip ; ./build_ub/x3v2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
  const float* A;            // [M][lda] fp32
  const unsigned short* Ap;  // [3][K/16][M][16] bf16 planes of A (APL)
  const unsigned short* Wp;  // [3][K/16][N][16] bf16 planes of W
  const float* bias;
  float* C;
  int lda, ldc, M, N, K, tiles_m, tiles_n;
  long long a_plane, w_plane;  // elements between planes
};

__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const bf16x2 b1 = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
  p1 = __builtin_bit_cast(unsigned, b1);
  const float r0 = x0 - __builtin_bit_cast(float, p1 << 16);
  const float r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
  const bf16x2 b2 = __builtin_convertvector(f32x2{r0, r1}, bf16x2);
  p2 = __builtin_bit_cast(unsigned, b2);
  const float s0 = r0 - __builtin_bit_cast(float, p2 << 16);
  const float s1 = r1 - __builtin_bit_cast(float, p2 & 0xffff0000u);
  const bf16x2 b3 = __builtin_convertvector(f32x2{s0, s1}, bf16x2);
  p3 = __builtin_bit_cast(unsigned, b3);
}

__global__ void split_planes_kernel(const float* __restrict__ W, int ldw, int N, int K, unsigned short* __restrict__ P) {
  const size_t n2 = (size_t)N * (K / 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (K / 2), c = (i % (K / 2)) * 2;
    unsigned p1, p2, p3;
    split2(W[row * ldw + c], W[row * ldw + c + 1], p1, p2, p3);
    unsigned* o = reinterpret_cast<unsigned*>(P);
    const size_t plane = (size_t)N * K / 2;
    const size_t at = ((c >> 4) * (size_t)N + row) * 8 + ((c & 15) >> 1);
    o[at] = p1; o[plane + at] = p2; o[2 * plane + at] = p3;
  }
}

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// one ds_read_b128 the compiler does not know about (it would put a vmcnt(0) in front of every LDS access that may
// alias an LDS-DMA destination); results are fenced by the explicit lgkmcnt waits below
template <int OFF = 0>
__device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}

constexpr int BM = 128, BN = 128, BK = 16;

// W fragment r = plane * 4 + ni of a slot whose W region starts at `addr` (+ the lane's fragment offset)
template <int R>
__device__ __forceinline__ void read_w(u32x4 (&wf)[3][4], unsigned addr) {
  wf[R >> 2][R & 3] = lds_read16<(R >> 2) * 4096 + (R & 3) * 1024>(addr);
}
template <int P>
__device__ __forceinline__ void read_apl(u32x4 (&af)[3], unsigned addr) { af[P] = lds_read16<P * 4096>(addr); }
// the same with the index as a (compile-time after unrolling) value: folds to one read
__device__ __forceinline__ void read_w_i(u32x4 (&wf)[3][4], unsigned addr, int r) {
  switch (r) {
    case 0: read_w<0>(wf, addr); break; case 1: read_w<1>(wf, addr); break; case 2: read_w<2>(wf, addr); break;
    case 3: read_w<3>(wf, addr); break; case 4: read_w<4>(wf, addr); break; case 5: read_w<5>(wf, addr); break;
    case 6: read_w<6>(wf, addr); break; case 7: read_w<7>(wf, addr); break; case 8: read_w<8>(wf, addr); break;
    case 9: read_w<9>(wf, addr); break; case 10: read_w<10>(wf, addr); break; default: read_w<11>(wf, addr); break;
  }
}
__device__ __forceinline__ void read_apl_i(u32x4 (&af)[3], unsigned addr, int p) {
  switch (p) { case 0: read_apl<0>(af, addr); break; case 1: read_apl<1>(af, addr); break; default: read_apl<2>(af, addr); break; }
}

// residual of a bf16 term, as plain (unpacked) VALU: the compiler's SLP pass would pair these into v_pk_add_f32, which
// issues slower than two v_sub_f32 beside MFMAs
__device__ __forceinline__ float fsub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <bool APL>
__global__ __launch_bounds__(256, 2) void x3v2_kernel(Args g) {
  constexpr int A_REG = APL ? 3 * BM * 32 : BM * 64;
  constexpr int W_REG = 3 * BN * 32;
  constexpr int SLOT = A_REG + W_REG;
  constexpr int NPA = APL ? 3 : 2;       // A pieces per wave and slice
  constexpr int NP = NPA + 3;            // DMA pieces per wave and slice
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsl = g.K / BK;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
  if (my_tiles <= 0) return;
  const int total = my_tiles * nsl;

  // ---- loader state: uniform base pointers (advance per slice) + per-lane BYTE offsets (change per tile) ----
  int ld_t = 0, ld_j = 0;
  unsigned a_off[2], w_off;
  const char* a_base = reinterpret_cast<const char*>(APL ? (const void*)g.Ap : (const void*)g.A);
  const char* w_base = reinterpret_cast<const char*>(g.Wp);
  const size_t a_step = APL ? (size_t)g.M * 32 : (size_t)BK * 4, w_step = (size_t)g.N * 32;
  const size_t a_pl = (size_t)g.a_plane * 2, w_pl = (size_t)g.w_plane * 2;
  auto set_tile = [&](int t) {
    const int id = blockIdx.x + t * G;
    const int m0 = (id / g.tiles_n) * BM, n0 = (id % g.tiles_n) * BN;
    if (APL) {
      const int lr = 32 * wave + (lane >> 1);
      int row = m0 + lr;
      row = row < g.M ? row : g.M - 1;
      a_off[0] = (unsigned)row * 32 + (((lane & 1) ^ ((lr >> 4) & 1)) * 16);
      a_off[1] = 0;
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int lr = 32 * wave + 16 * q + (lane >> 2);
        int row = m0 + lr;
        row = row < g.M ? row : g.M - 1;
        a_off[q] = ((unsigned)row * g.lda + 4 * ((lane & 3) ^ ((lr >> 2) & 3))) * 4;
      }
    }
    {
      const int lr = 32 * wave + (lane >> 1);
      int row = n0 + lr;
      row = row < g.N ? row : g.N - 1;
      w_off = (unsigned)row * 32 + (((lane & 1) ^ ((lr >> 4) & 1)) * 16);
    }
  };
  // piece k of the slice the loader stands on -> ring slot `slot`
  auto issue_piece = [&](int k, int slot) {
    unsigned char* base = lds + slot * SLOT;
    if (k < NPA) {
      if (APL) __builtin_amdgcn_global_load_lds(a_base + k * a_pl + a_off[0], LDS_PTR(base + k * (BM * 32) + wave * 1024), 16, 0, 0);
      else __builtin_amdgcn_global_load_lds(a_base + a_off[k], LDS_PTR(base + (32 * wave + 16 * k) * 64), 16, 0, 0);
    } else {
      const int p = k - NPA;
      __builtin_amdgcn_global_load_lds(w_base + p * w_pl + w_off, LDS_PTR(base + A_REG + p * (BN * 32) + wave * 1024), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++ld_j == nsl) {
      if (ld_t + 1 < my_tiles) {
        ld_j = 0; set_tile(++ld_t);
        a_base -= (size_t)(nsl - 1) * a_step; w_base -= (size_t)(nsl - 1) * w_step;
      } else ld_j = nsl - 1;   // past the end: the last slice again (never read)
    } else { a_base += a_step; w_base += w_step; }
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NP; ++k) issue_piece(k, slot);
    advance();
  };

  // ---- fragment addresses (bytes inside a slot) ----
  const unsigned fw = A_REG + l32 * 32 + ((half ^ (l32 >> 4)) * 16);                       // + p*4096 + ni*1024
  const unsigned fa_pl = (32 * wave + l32) * 32 + ((half ^ (l32 >> 4)) * 16);              // planes: + p*4096
  const unsigned fa_r0 = (32 * wave + l32) * 64 + (((2 * half) ^ ((l32 >> 2) & 3)) * 16);  // fp32: k 8*half .. +3
  const unsigned fa_r1 = (32 * wave + l32) * 64 + (((2 * half + 1) ^ ((l32 >> 2) & 3)) * 16);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;

  u32x4 wf[2][3][4];   // [set][plane][ni]
  u32x4 af[2][3];      // [set][plane]
  u32x4 ar[2];         // fp32 rows of the next slice (8 floats)
  f32x16 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;

  // split of the 8 floats in ar[] in eight steps (pair q = step / 2, level = step % 2) so that the VALU work can be dealt
  // out over the MFMA gaps; the results are pinned where they are computed (the optimiser would sink them to their use)
  float r_[4][2];
  unsigned p1_[4], p2_[4], p3_[4];
  auto split_step = [&](int st) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int q = st >> 1;
    if ((st & 1) == 0) {
      const f32x4 v = __builtin_bit_cast(f32x4, ar[q >> 1]);
      const float x0 = v[2 * (q & 1)], x1 = v[2 * (q & 1) + 1];
      p1_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
      r_[q][0] = fsub(x0, __builtin_bit_cast(float, p1_[q] << 16));
      r_[q][1] = fsub(x1, __builtin_bit_cast(float, p1_[q] & 0xffff0000u));
      asm volatile("" : "+v"(p1_[q]), "+v"(r_[q][0]), "+v"(r_[q][1]));
    } else {
      p2_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r_[q][0], r_[q][1]}, bf16x2));
      const float s0 = fsub(r_[q][0], __builtin_bit_cast(float, p2_[q] << 16));
      const float s1 = fsub(r_[q][1], __builtin_bit_cast(float, p2_[q] & 0xffff0000u));
      p3_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
      asm volatile("" : "+v"(p2_[q]), "+v"(p3_[q]));
    }
  };
  auto split_collect = [&](u32x4 (&dst)[3]) {
    dst[0] = u32x4{p1_[0], p1_[1], p1_[2], p1_[3]};
    dst[1] = u32x4{p2_[0], p2_[1], p2_[2], p2_[3]};
    dst[2] = u32x4{p3_[0], p3_[1], p3_[2], p3_[3]};
  };

  // ---- compute-side state ----
  int cp_t = 0, cp_j = 0;
  auto epilogue = [&]() {
    const int id = blockIdx.x + cp_t * G;
    const int m0 = (id / g.tiles_n) * BM, n0 = (id % g.tiles_n) * BN;
    const int m = m0 + 32 * wave + l32;
    f32x4 out[16];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int n = n0 + ni * 32 + 8 * q + 4 * half;
        asm volatile("" : "+v"(n));
        const f32x4 b = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        out[ni * 4 + q] = f32x4{acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]} + b;
        acc[ni][4 * q] = 0.f; acc[ni][4 * q + 1] = 0.f; acc[ni][4 * q + 2] = 0.f; acc[ni][4 * q + 3] = 0.f;
      }
    if (m < g.M) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(g.C + (size_t)m * g.ldc + n0 + ni * 32 + 8 * q + 4 * half) = out[ni * 4 + q];
    }
    // drain the vector-memory counter on this rare path: merged into the K loop, the unknown state left here would
    // open every slice with s_waitcnt vmcnt(0)
    __builtin_amdgcn_s_waitcnt(0x0F70);
  };

  // ---- prologue: slices 0, 1, 2 in flight; 0 and 1 landed; fragments of slice 0 in set 0 ----
  set_tile(0);
  issue(0); issue(1); issue(2);
  __builtin_amdgcn_s_waitcnt(0x0F70 | NP);   // vmcnt(NP): all but the youngest slice
  __builtin_amdgcn_s_barrier();
  {
    if (APL) {
#pragma unroll
      for (int p = 0; p < 3; ++p) af[0][p] = lds_read16(lds0 + fa_pl + p * 4096);
    } else {
      ar[0] = lds_read16(lds0 + fa_r0);
      ar[1] = lds_read16(lds0 + fa_r1);
    }
    read_w<0>(wf[0], lds0 + fw); read_w<1>(wf[0], lds0 + fw); read_w<2>(wf[0], lds0 + fw); read_w<3>(wf[0], lds0 + fw);
    read_w<4>(wf[0], lds0 + fw); read_w<5>(wf[0], lds0 + fw); read_w<6>(wf[0], lds0 + fw); read_w<7>(wf[0], lds0 + fw);
    read_w<8>(wf[0], lds0 + fw); read_w<9>(wf[0], lds0 + fw); read_w<10>(wf[0], lds0 + fw); read_w<11>(wf[0], lds0 + fw);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0]), "+v"(ar[1])::"memory");
    if (!APL) {
#pragma unroll
      for (int st = 0; st < 8; ++st) split_step(st);
      split_collect(af[0]);
    }
  }
  __builtin_amdgcn_s_barrier();

  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};   // (A plane, W plane), small terms first
  constexpr int NRD = APL ? 15 : 14;
  int s0 = 0, s1 = 1;   // ring slots of slice s, s + 1  (slice s + 3 goes to slot s0)
  for (int s = 0; s < total; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned nb = lds0 + s1 * SLOT;   // slot of slice s + 1
      // 24 MFMAs of slice s; in their gaps: the reads of slice s + 1 (one per gap), the split of its rows, the DMA of s + 3
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        const int t = i >> 2, ni = i & 3;
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[u][PB[t]][ni]),
                                                          __builtin_bit_cast(bf16x8, af[u][PA[t]]), acc[ni], 0, 0, 0);
        if (APL) {
          if (i < 3) read_apl_i(af[u ^ 1], nb + fa_pl, i);
          else if (i < 15) read_w_i(wf[u ^ 1], nb + fw, i - 3);
        } else {
          if (i == 0) ar[0] = lds_read16(nb + fa_r0);
          else if (i == 1) ar[1] = lds_read16(nb + fa_r1);
          else if (i < 14) read_w_i(wf[u ^ 1], nb + fw, i - 2);
          if (i == 5) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ar[0]), "+v"(ar[1])::"memory");   // the two row reads are the oldest of six
          if (i >= 5 && i < 13) split_step(i - 5);
          if (i == 13) split_collect(af[u ^ 1]);
        }
        if (i >= NRD && i < NRD + NP) issue_piece(i - NRD, s0);
        if (i == NRD + NP) advance();
        __builtin_amdgcn_sched_barrier(0);
      }
      // slice s + 2 has landed (own pieces), every fragment of slice s + 1 is in registers
      __builtin_amdgcn_s_waitcnt(0x0070 | NP);   // vmcnt(NP) lgkmcnt(0)
      asm volatile("" : "+v"(wf[u ^ 1][0][0]), "+v"(wf[u ^ 1][0][1]), "+v"(wf[u ^ 1][0][2]), "+v"(wf[u ^ 1][0][3]),
                        "+v"(wf[u ^ 1][1][0]), "+v"(wf[u ^ 1][1][1]), "+v"(wf[u ^ 1][1][2]), "+v"(wf[u ^ 1][1][3]),
                        "+v"(wf[u ^ 1][2][0]), "+v"(wf[u ^ 1][2][1]), "+v"(wf[u ^ 1][2][2]), "+v"(wf[u ^ 1][2][3]));
      if (APL) asm volatile("" : "+v"(af[u ^ 1][0]), "+v"(af[u ^ 1][1]), "+v"(af[u ^ 1][2]));
      if (++cp_j == nsl) { epilogue(); cp_j = 0; ++cp_t; }
      __builtin_amdgcn_s_barrier();
      { const int tmp = s0; s0 = s1; s1 = 3 - s0 - tmp; }   // (s0, s1, s2) -> (s1, s2, s0)
    }
  }
}

// ---- fp64 reference on a sample of rows ----
__global__ void ref_kernel(const float* A, int lda, const float* W, const float* bias, double* R, int M, int N, int K, int row_step) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int mi = blockIdx.y;
  const int m = mi * row_step;
  if (n >= N || m >= M) return;
  double s = bias ? bias[n] : 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(size_t)m * lda + k] * (double)W[(size_t)n * K + k];
  R[(size_t)mi * N + n] = s;
}

template <bool APL>
double run(const Args& g, int iters) {
  constexpr int A_REG = APL ? 3 * BM * 32 : BM * 64;
  const int bytes = 3 * (A_REG + 3 * BN * 32);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&x3v2_kernel<APL>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  const int tiles = g.tiles_m * g.tiles_n;
  const int grid = tiles < 512 ? tiles : 512;
  hipLaunchKernelGGL(x3v2_kernel<APL>, dim3(grid), dim3(256), bytes, 0, g);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(x3v2_kernel<APL>, dim3(grid), dim3(256), bytes, 0, g);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e-3;
}

int main(int argc, char** argv) {
  const bool zeros = argc > 1 && !strcmp(argv[1], "--zeros");   // issue-bound rate (the chip clocks to its power budget)
  if (zeros) printf("operands: zeros\n");
  const int Ms[] = {2048, 4096, 9216, 16384, 32768};
  const int shapes[4][2] = {{512, 1536}, {512, 512}, {512, 1024}, {1024, 512}};   // K, N
  printf("%8s %5s %5s | %12s %12s | %10s %10s\n", "M", "K", "N", "fp32-A TF/s", "planes TF/s", "err fp32A", "err planes");
  for (int M : Ms)
    for (auto& sh : shapes) {
      const int K = sh[0], N = sh[1];
      std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N);
      srand(1);
      for (auto& v : hA) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
      for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
      for (auto& v : hb) v = (rand() / (float)RAND_MAX - 0.5f);
      if (zeros) { for (auto& v : hA) v = 0.f; for (auto& v : hW) v = 0.f; }
      float *A, *W, *b, *C;
      unsigned short *Ap, *Wp;
      CHECK(hipMalloc(&A, hA.size() * 4)); CHECK(hipMalloc(&W, hW.size() * 4)); CHECK(hipMalloc(&b, N * 4));
      CHECK(hipMalloc(&C, (size_t)M * N * 4)); CHECK(hipMalloc(&Ap, hA.size() * 6)); CHECK(hipMalloc(&Wp, hW.size() * 6));
      CHECK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
      CHECK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(split_planes_kernel, dim3(1024), dim3(256), 0, 0, A, K, M, K, Ap);
      hipLaunchKernelGGL(split_planes_kernel, dim3(1024), dim3(256), 0, 0, W, K, N, K, Wp);
      Args g{A, Ap, Wp, b, C, K, N, M, N, K, (M + BM - 1) / BM, (N + BN - 1) / BN, (long long)M * K, (long long)N * K};
      const int row_step = M / 64;
      double* R;
      CHECK(hipMalloc(&R, (size_t)64 * N * 8));
      hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, 64), dim3(256), 0, 0, A, K, W, b, R, M, N, K, row_step);
      std::vector<double> hR((size_t)64 * N);
      CHECK(hipMemcpy(hR.data(), R, hR.size() * 8, hipMemcpyDeviceToHost));
      const double flops = 2.0 * M * N * K;
      const int iters = flops > 2e10 ? 20 : 50;
      double tf[2], err[2];
      std::vector<float> hC((size_t)M * N);
      for (int v = 0; v < 2; ++v) {
        CHECK(hipMemset(C, 0xff, (size_t)M * N * 4));
        const double dt = v == 0 ? run<false>(g, iters) : run<true>(g, iters);
        tf[v] = flops / dt / 1e12;
        CHECK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double e = 0, sc = 0;
        for (int mi = 0; mi < 64; ++mi)
          for (int n = 0; n < N; ++n) {
            const double r = hR[(size_t)mi * N + n], c = hC[(size_t)(mi * row_step) * N + n];
            e = fmax(e, fabs(r - c)); sc = fmax(sc, fabs(r));
          }
        err[v] = e / sc;
      }
      printf("%8d %5d %5d | %12.1f %12.1f | %10.2e %10.2e\n", M, K, N, tf[0], tf[1], err[0], err[1]);
      hipFree(A); hipFree(W); hipFree(b); hipFree(C); hipFree(Ap); hipFree(Wp); hipFree(R);
    }
  return 0;
}
