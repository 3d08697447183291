// fp32-accurate dense projection on the bf16 matrix cores ("3 x bf16"):
//     C = act(Asel * W^T + bias) + residual,   A fp32 [M,K], W fp32 [N,K] given as three bf16 planes.
//
// Every fp32 value is split EXACTLY into three bf16 terms, x = x1 + x2 + x3 (round-to-nearest each
// time, |x - x1 - x2 - x3| <= 2^-25 |x|), and a product is evaluated as the six partial products whose
// weight is >= 2^-16:  x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1); what is dropped is <= 2^-24 |xy|,
// the size of one fp32 rounding.  bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result carries the error of an ordinary fp32 dot product (measured against
// fp64: tools/ubench/gemm_bf16x3.hip, tests/test_hip_ops.py::test_gemm_x3_*) while the matrix pipe runs
// at 2.5 PF/s / 6 = 417 TF/s fp32-equivalent instead of the 157 TF/s of v_mfma_f32_32x32x2_f32.
//
// Plane layout: [plane][k / 16][row][k % 16].  A K-slice of 16 is then 32 contiguous bytes per row AND
// contiguous over rows: a staging instruction that covers 32 rows reads one aligned 1 KB block (eight full
// 128-byte lines).  With row-major [row][K] planes the same instruction touched 32 lines for 32 bytes each
// and the kernel ran at the L2 -> CU line bandwidth (4x the bytes it needed; measured 121 -> 192 TF/s
// with the loads removed).
//
// Weights are split once (ff_split_weight_bf16x3, at model bind time); activations are split on the
// fly while they are staged into LDS (44 VALU ops per thread and 16-wide K slice, hidden under the 24
// MFMAs of the slice).
//
// Kernel: the stream-K flat (tile, slice) pipeline of ff_gemm.hip's gemm_streamk_kernel with
//   * 128x128 block tiles, 4 waves x (64x64) = four 32x32 accumulators per wave, K slices of 16;
//   * LDS: [3-slot ring][3 planes][256 rows][32 B], 16-byte chunk c of row r stored at c ^ ((r >> 4) & 1)
//     (ds_read_b128 serves lanes {0-3,12-15,20-27} / {4-11,16-19,28-31} together over 64 banks: rows 8 apart
//     share a bank octet and must use different halves; no padding needed): 72 KB -> two blocks per CU;
//   * per slice and wave: 12 ds_read_b128 (this slice's fragments; the co-resident block covers their
//     latency -- a second fragment set does not fit in 256 VGPRs), 6 ds_write_b64 + 3 ds_write_b128
//     (slice +2), 5 global loads (slice +4), 24 MFMAs, one barrier;
//   * equal ranges of 32-wide K units per block, partial tiles (64 KB) handed over through sc1 accesses
//     and summed by the owning block in ascending block order (deterministic).
#include <mutex>
#include <vector>

#include "ff_common.h"
#include "ff_device.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct X3Args {
  const float* A;
  const float* A2;
  const unsigned short* Wp;  // [3][K/16][N][16] bf16 (see "plane layout" above)
  const float* bias;
  const float* res;
  float* C;
  int lda, ldr, ldc;
  int M, N, K;
  int n_split, act;
  int tiles_m, tiles_n;
  long long plane_stride;  // elements between two planes of Wp
  float* ws;               // [grid][64 accumulator registers][256 threads]
  unsigned int* flags;     // [grid]: 1 = slot holds a partial tile
  int upt;                 // units (32 k) per tile
  int base, rem;           // block lb owns base + (lb < rem) units
};

// x (two floats) -> packed bf16 pairs of the three terms
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

__global__ void split_weight_kernel(const float* __restrict__ W, int ldw, int N, int K,
                                    unsigned short* __restrict__ P) {
  const size_t n2 = (size_t)N * (K / 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (K / 2), c = (i % (K / 2)) * 2;
    unsigned p1, p2, p3;
    split2(W[row * ldw + c], W[row * ldw + c + 1], p1, p2, p3);
    unsigned* o = reinterpret_cast<unsigned*>(P);
    const size_t plane = (size_t)N * K / 2;
    const size_t at = ((c >> 4) * (size_t)N + row) * 8 + ((c & 15) >> 1);  // [k/16][row][16] in bf16 pairs
    o[at] = p1;
    o[plane + at] = p2;
    o[2 * plane + at] = p3;
  }
}

// Measured and dropped: one block of eight waves per CU on a 128x256 tile with a four-slot ring (three
// slices of DMA lead, 25 % fewer operand bytes per MFMA): 132 / 125 / 124 TF/s where the four-wave kernel
// below reaches 164 / 154 / 146 -- the eight-wave barrier domain costs more than the extra lead buys.
//
// End of a segment of the flat (tile, slice) sequence: hand the partial tile over (kind 1), or finish the
// tile -- after adding the partials of the lower-numbered blocks (kind 2) -- with bias / activation /
// residual and the store.
__device__ __forceinline__ void x3_end_segment(const X3Args& g, f32x16 (&acc)[2][2], int cp_kind, int lb, int k0,
                                               int e_m0, int e_n0) {
  constexpr int BM = 128, BN = 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int upt = g.upt;
  {
    if (cp_kind == 1) {  // hand over the raw accumulators: slot[lb][64][tid]
      // slot[lb][16 quads][256 threads] float4, written / read with 16-byte sc1 accesses (inline asm: the
      // compiler has no vector form of an agent-coherent access; the loads are fenced by the explicit wait)
      f32x4* wp = reinterpret_cast<f32x4*>(g.ws + (size_t)lb * (BM * BN)) + tid;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = {acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]};
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + ((mi * 2 + ni) * 4 + q) * 256), "v"(v) : "memory");
            acc[mi][ni][4 * q] = 0.f; acc[mi][ni][4 * q + 1] = 0.f; acc[mi][ni][4 * q + 2] = 0.f; acc[mi][ni][4 * q + 3] = 0.f;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(g.flags + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (cp_kind == 2) {  // add the partials of the blocks that hold units [k0 * upt, u0) of this tile
      const int ub = k0 * upt;
      const int big = g.rem * (g.base + 1);
      const int c0 = ub < big ? ub / (g.base + 1) : g.rem + (ub - big) / g.base;
      for (int c = c0; c < lb; ++c) {
        while (__hip_atomic_load(g.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const f32x4* rp = reinterpret_cast<const f32x4*>(g.ws + (size_t)c * (BM * BN)) + tid;
        f32x4 t[16];  // all 16 loads in flight before the first use: one memory round trip per contributor
#pragma unroll
        for (int q = 0; q < 16; ++q)
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[q]) : "v"(rp + q * 256) : "memory");
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]),
                       "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] += t[(mi * 2 + ni) * 4 + (e >> 2)][e & 3];
      }
      __syncthreads();  // every thread is past its flag polls
      if (tid < lb - c0) __hip_atomic_store(g.flags + c0 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // bias, activation, residual, store.  Per 32-column half of the wave tile: the residual values of both 32x32 sub-tiles
    // are requested together, every result is finished in its accumulator register, and only then the 32 stores go out back
    // to back.  (Stores inside the per-element loop were each preceded by `s_waitcnt vmcnt(0)` -- the compiler re-establishes
    // "the residual has arrived" in every guarded block and on gfx9 that counter also counts the store before: 64 serialised
    // write round trips per tile, about as long as the tile's whole K = 512 MFMA chain; see gemm_persist_body.)  A lane reads
    // and writes the same elements, so a residual that aliases C stays correct.
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      int col = e_n0 + wn0 + ni * 32 + l32;
      asm volatile("" : "+v"(col));  // opaque: the address arithmetic below must not be hoisted out of the K loop
      const bool colok = col < g.N;
      const int colc = colok ? col : g.N - 1;
      const float bv = g.bias ? g.bias[colc] : 0.f;
      int rbase[2];
      float rl[2][16];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        rbase[mi] = e_m0 + wm0 + mi * 32 + 4 * half;
        asm volatile("" : "+v"(rbase[mi]));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int row = rbase[mi] + (e & 3) + 8 * (e >> 2);
          row = row < g.M ? row : g.M - 1;
          rl[mi][e] = g.res ? g.res[(size_t)row * g.ldr + colc] : 0.f;
        }
      }
      float fin[2][16];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[mi][ni][e] + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          fin[mi][e] = v + rl[mi][e];
          acc[mi][ni][e] = 0.f;
        }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) ff_store_tile<false>(g.C, g.ldc, rbase[mi], col, g.M, colok, fin[mi]);
    }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_x3_kernel(X3Args g) {
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int PLANE_B = (BM + BN) * 32, BUF_B = 3 * PLANE_B;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // ---- this block's unit range and its segments (see gemm_streamk_kernel) ----
  const int G = gridDim.x;
  const int lb = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int upt = g.upt;
  const int u0 = lb * g.base + (lb < g.rem ? lb : g.rem);
  const int u1 = u0 + g.base + (lb < g.rem ? 1 : 0);
  if (u0 >= u1) return;
  const int k0 = u0 / upt, k1 = (u1 - 1) / upt;
  const int ja = u0 - k0 * upt;
  const int jb = u1 - k1 * upt;
  const bool has_c = jb < upt;
  const bool has_o = ja > 0 && !(k0 == k1 && has_c);
  const int kf0 = k0 + (ja > 0 ? 1 : 0);
  const int nfull = (k1 + (has_c ? 0 : 1) - kf0) > 0 ? (k1 + (has_c ? 0 : 1) - kf0) : 0;
  const int nseg = (has_c ? 1 : 0) + nfull + (has_o ? 1 : 0);
  auto segment = [&](int p, int& tile, int& j0, int& n, int& kind) {
    if (has_c && p == 0) {
      tile = k1; j0 = 2 * (k1 == k0 ? ja : 0); n = 2 * jb - j0; kind = 1;
    } else {
      const int q = p - (has_c ? 1 : 0);
      if (q < nfull) { tile = kf0 + q; j0 = 0; n = nsl; kind = 0; }
      else { tile = k0; j0 = 2 * ja; n = nsl - j0; kind = 2; }
    }
  };

  // ---- staging maps ----
  // A (fp32): thread (ar, ac) loads 4 floats of rows ar, ar + 64; written as 3 x 8 bytes per row
  const int ar = tid >> 2, ac = tid & 3;
  // W (bf16 planes): thread (wr, wh) loads 8 bf16 of row wr in each of the three planes
  const int wr = tid >> 1, wh = tid & 1;
  const float* a_ptr[2];
  const unsigned short* w_ptr[3];  // per plane, advanced by one K block per slice
  const long long w_step = (long long)g.N * 16;
  auto set_load_tile = [&](int id, int j0) {
    const int rem2 = id % tiles_mn;
    const int m0 = (rem2 / g.tiles_n) * BM, n0 = (rem2 % g.tiles_n) * BN;
    const float* Asrc = (g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int row = m0 + ar + 64 * p;
      row = row < g.M ? row : g.M - 1;
      a_ptr[p] = Asrc + (size_t)row * g.lda + ac * 4;
    }
    int n = n0 + wr;
    n = n < g.N ? n : g.N - 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) w_ptr[q] = g.Wp + q * g.plane_stride + j0 * w_step + (size_t)n * 16 + wh * 8;
  };
  int ld_p = 0, ld_j, ld_end;
  {
    int tile, j0, n, kind;
    segment(0, tile, j0, n, kind);
    set_load_tile(tile, j0);
    ld_j = j0; ld_end = j0 + n;
  }
  f32x4 sa[2][2];
  u32x4 sw[2][3];
  auto load_next = [&](f32x4* xa, u32x4* xw) {
    const int kk0 = ld_j * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) xa[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + kk0);
#pragma unroll
    for (int q = 0; q < 3; ++q) xw[q] = *reinterpret_cast<const u32x4*>(w_ptr[q]);
  };
  auto advance = [&]() {
    if (++ld_j == ld_end) {
      if (ld_p + 1 < nseg) {
        int tile, j0, n, kind;
        segment(++ld_p, tile, j0, n, kind);
        set_load_tile(tile, j0);
        ld_j = j0; ld_end = j0 + n;
      } else {
        ld_j = ld_end - 1;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q) w_ptr[q] += w_step;
    }
  };
  // LDS addresses (bytes inside a ring slot)
  int sta[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int row = ar + 64 * p;
    sta[p] = row * 32 + (((ac >> 1) ^ ((row >> 4) & 1)) * 16) + (ac & 1) * 8;
  }
  const int stw = (BM + wr) * 32 + ((wh ^ ((wr >> 4) & 1)) * 16);
  auto store_from = [&](const f32x4* xa, const u32x4* xw, int buf) {
    unsigned char* base = lds + buf * BUF_B;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      unsigned a1, a2, a3, b1, b2, b3;
      split2(xa[p][0], xa[p][1], a1, a2, a3);
      split2(xa[p][2], xa[p][3], b1, b2, b3);
      *reinterpret_cast<u32x2*>(base + sta[p]) = u32x2{a1, b1};
      *reinterpret_cast<u32x2*>(base + PLANE_B + sta[p]) = u32x2{a2, b2};
      *reinterpret_cast<u32x2*>(base + 2 * PLANE_B + sta[p]) = u32x2{a3, b3};
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(base + q * PLANE_B + stw) = xw[q];
  };
  // fragment addresses: the swizzle bit of rows wm0 + mi*32 + l32 only depends on l32
  const int fchunk = (half ^ ((l32 >> 4) & 1)) * 16;
  const int fra = (wm0 + l32) * 32 + fchunk;
  const int frb = (BM + wn0 + l32) * 32 + fchunk;
  bf16x8 fa[3][2], fb[3][2];
  auto read_frags = [&](bf16x8 (*xa)[2], bf16x8 (*xb)[2], int buf) {
    const unsigned char* base = lds + buf * BUF_B;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xa[p][i] = *reinterpret_cast<const bf16x8*>(base + p * PLANE_B + fra + i * 32 * 32);
        xb[p][i] = *reinterpret_cast<const bf16x8*>(base + p * PLANE_B + frb + i * 32 * 32);
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  auto mfma_frags = [&](bf16x8 (*xa)[2], bf16x8 (*xb)[2]) {
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};  // small terms first
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[PA[t]][mi], xb[PB[t]][ni], acc[mi][ni], 0, 0, 0);
  };

  // ---- compute-side segment state ----
  int cp_p = 0, cp_cnt = 0, cp_n = 0, cp_kind = 0;
  int e_m0 = 0, e_n0 = 0;
  auto begin_segment = [&](int p) {
    int id, j0;
    segment(p, id, j0, cp_n, cp_kind);
    cp_cnt = 0;
    const int rem2 = id % tiles_mn;
    e_m0 = (rem2 / g.tiles_n) * BM;
    e_n0 = (rem2 % g.tiles_n) * BN;
  };
  auto end_segment = [&]() { x3_end_segment(g, acc, cp_kind, lb, k0, e_m0, e_n0); };

  // ---- prologue: slices 0,1 -> LDS; slices 2,3 -> staging registers ----
  load_next(sa[0], sw[0]); advance();
  load_next(sa[1], sw[1]); advance();
  store_from(sa[0], sw[0], 0);
  store_from(sa[1], sw[1], 1);
  load_next(sa[0], sw[0]); advance();
  load_next(sa[1], sw[1]); advance();
  begin_segment(0);
  __syncthreads();

  int b0 = 0, b1 = 1, b2 = 2;
  const int total_slices = 2 * (u1 - u0);
  for (int s = 0; s < total_slices; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      read_frags(fa, fb, b0);
      store_from(sa[u], sw[u], b2);
      load_next(sa[u], sw[u]);
      mfma_frags(fa, fb);
      advance();
      if (++cp_cnt == cp_n) {  // block-uniform: the last slice of the segment was just issued
        end_segment();
        if (++cp_p < nseg) begin_segment(cp_p);
        // Drain the vector-memory counter on this (rare) path: otherwise the compiler's wait-count
        // analysis merges the unknown state left by the loops above into the K loop and opens every
        // slice with s_waitcnt vmcnt(0), i.e. exposes the full latency of the loads issued one slice ago.
        __builtin_amdgcn_s_waitcnt(0x0F70);
      }
      __syncthreads();
      { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
    }
  }
}

// Partial-tile workspace: one per (device, stream), as in ff_gemm.hip (64 KB slots here).
constexpr int X3_MAX_GRID = 512;
struct X3Workspace {
  int device;
  hipStream_t st;
  float* ws;
  unsigned int* flags;
};
std::mutex g_x3_mu;
std::vector<X3Workspace> g_x3;

int x3_acquire(hipStream_t st, X3Args* out) {
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_x3_mu);
  for (X3Workspace& w : g_x3)
    if (w.device == dev && w.st == st) {
      out->ws = w.ws; out->flags = w.flags;
      return FF_OK;
    }
  X3Workspace w{dev, st, nullptr, nullptr};
  FF_CHECK_HIP(hipMalloc(&w.ws, (size_t)X3_MAX_GRID * 128 * 128 * sizeof(float)));
  FF_CHECK_HIP(hipMalloc(&w.flags, X3_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipMemset(w.flags, 0, X3_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipDeviceSynchronize());
  g_x3.push_back(w);
  out->ws = w.ws; out->flags = w.flags;
  return FF_OK;
}

}  // namespace

extern "C" size_t ff_split_weight_bytes(int N, int K) { return (size_t)3 * N * K * sizeof(unsigned short); }

extern "C" int ff_split_weight_bf16x3(const float* W, int ldw, int N, int K, void* planes, ff_stream_t stream) {
  FF_CHECK_ARG(W && planes && N > 0 && K > 0 && (K & 15) == 0 && ldw >= K, "ff_split_weight_bf16x3: bad arguments (K %% 16)");
  FF_CHECK_ARG(ff_aligned16(planes), "ff_split_weight_bf16x3: planes must be 16-byte aligned");
  const size_t n2 = (size_t)N * (K / 2);
  const int grid = (int)((n2 + 255) / 256 < 4096 ? (n2 + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     static_cast<unsigned short*>(planes));
  FF_CHECK_LAUNCH();
  return FF_OK;
}

extern "C" int ff_x3_prepare_stream(hipStream_t st) {
  X3Args g;
  return x3_acquire(st, &g);
}

extern "C" int ff_gemm_x3(const float* A, int lda, const float* A2, int n_split, const void* w_planes,
                          const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N,
                          int K, int act, ff_stream_t stream) {
  if (M == 0 || N == 0) return FF_OK;
  FF_CHECK_ARG(M > 0 && N > 0 && K >= 64 && (K % 32) == 0, "ff_gemm_x3: bad M=%d N=%d K=%d (K %% 32, K >= 64)", M, N, K);
  FF_CHECK_ARG(A && w_planes && C, "ff_gemm_x3: null operand");
  FF_CHECK_ARG(ff_aligned16(w_planes) && ((size_t)N * K % 8) == 0, "ff_gemm_x3: weight planes must be 16-byte aligned");
  FF_CHECK_ARG((lda & 3) == 0 && lda >= K && ff_aligned16(A) && (!A2 || ff_aligned16(A2)),
               "ff_gemm_x3: A/A2 must be 16-byte aligned with lda %% 4 == 0 (lda=%d)", lda);
  FF_CHECK_ARG(ldc >= N, "ff_gemm_x3: bad ldc");
  FF_CHECK_ARG(!residual || ldr >= N, "ff_gemm_x3: bad ldr");
  FF_CHECK_ARG(act == 0 || act == 1, "ff_gemm_x3: act must be 0 or 1");
  if (A2) FF_CHECK_ARG(n_split > 0 && n_split < N && (n_split % 128) == 0, "ff_gemm_x3: n_split must be a multiple of 128 inside (0,N)");
  hipStream_t st = (hipStream_t)stream;
  X3Args g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.A2 = A2; g.lda = lda;
  g.Wp = static_cast<const unsigned short*>(w_planes); g.bias = bias; g.res = residual; g.ldr = ldr;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.n_split = A2 ? n_split : N; g.act = act;
  g.tiles_m = ff_cdiv(M, 128); g.tiles_n = ff_cdiv(N, 128);
  g.plane_stride = (long long)N * K;
  g.upt = K / 32;
  const long units = (long)g.tiles_m * g.tiles_n * g.upt;
  FF_CHECK_ARG(units < (1L << 30), "ff_gemm_x3: problem too large");
  // Launch shape.  A cut tile costs its owner one memory round trip per contributing block (64 KB each),
  // so small launches cut every tile in two (four below 32 tiles) rather than into many pieces; from 256
  // tiles on, 512 blocks get equal unit ranges (a tile then spans at most three blocks).
  const long tiles = (long)g.tiles_m * g.tiles_n;
  long grid;
  if (tiles > X3_MAX_GRID / 2) grid = X3_MAX_GRID;
  else {
    long sf = tiles <= 32 ? 4 : 2;
    if (sf > g.upt) sf = g.upt;
    grid = tiles * sf;
    if (grid > X3_MAX_GRID) grid = X3_MAX_GRID;
  }
  if (grid > units) grid = units;
  g.base = (int)(units / grid);
  g.rem = (int)(units % grid);
  FF_RETURN_IF(x3_acquire(st, &g));
  static bool attr_set[16] = {};   // hipFuncSetAttribute is per device
  constexpr int bytes = 3 * 3 * 256 * 32;
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  FFProfScope prof(FF_CAT_GEMM, 2.0 * M * N * K, st);
  hipLaunchKernelGGL(gemm_x3_kernel, dim3((int)grid), dim3(256), bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
