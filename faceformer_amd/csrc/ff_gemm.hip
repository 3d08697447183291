// Dense fp32 projection on the CDNA4 matrix cores:  C = act(Asel * W^T + bias) + residual.
//
// v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles/SIMD, 157.3 TF/s chip peak): lane l supplies
// A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the accumulator lane layout is
// col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).  Both operands of our product are
// K-contiguous (activations [M,K], nn.Linear weights [N,K]), so a block stages a BM x 32 slice of A
// and a BN x 32 slice of W through LDS (row stride 36 floats: 16-byte aligned and conflict-free for
// ds_read_b128 / ds_write_b128) and each lane reads its 16 k-values (lane half 0: k 0..15, half 1:
// k 16..31 of the slice -- any k permutation is legal as long as A and W use the same one) with
// four 16-byte LDS reads per 32x32 sub-tile.  Global loads of slice t+1 are issued before the MFMA
// chain of slice t and written to the other LDS buffer after it (one barrier per slice).
//
// 256 threads = 4 waves (one per SIMD), each wave owns a WM x WN sub-tile (MI x NI accumulators).
// Blocks are remapped XCD-aware so that the column tiles of one A row-panel share an L2.
#include "ff_common.h"

namespace {

struct GemmArgs {
  const float* A;
  const float* A2;
  const float* W;
  const float* bias;
  const float* res;
  float* C;
  int lda, ldw, ldr, ldc;
  int M, N, K;
  int n_split, act;
  int tiles_m, tiles_n;
  long long batch_stride_a, batch_stride_w, batch_stride_c;  // blockIdx.y = batch index (elements)
};

template <int BM, int BN, int BK>
constexpr int gemm_lds_bytes() { return 2 * (BM + BN) * (BK + 4) * (int)sizeof(float); }

// ABL (timing ablations only, results are garbage): 1 = no global loads / LDS stores in the loop,
// 2 = no MFMA, 3 = no barriers in the loop.
template <int BM, int BN, int WM, int WN, int BK, int ABL = 0, int PF = 1>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int LDS_LD = BK + 4;        // 36 / 68 floats: 16-byte aligned, conflict-free b128 rows
  constexpr int TPR = BK / 4;           // threads (float4 columns) per row
  constexpr int RPP = 256 / TPR;        // rows per staging pass
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;
  static_assert(A_PASSES >= 1 && W_PASSES >= 1, "tile too small for this BK");
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int nblocks = g.tiles_m * g.tiles_n;
  const int lid = ff_xcd_remap(blockIdx.x, nblocks);
  const int m0 = (lid / g.tiles_n) * BM;
  const int n0 = (lid % g.tiles_n) * BN;
  const long long bz = blockIdx.y;
  const float* __restrict__ Asrc =
      ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  const float* __restrict__ W = g.W + bz * g.batch_stride_w;
  float* __restrict__ Cout = g.C + bz * g.batch_stride_c;

  // ---- global -> register staging (thread: float4 column c4 of rows r, r+32, ...) -------------
  const int c4 = tid % TPR, r = tid / TPR;
  f32x4 ra[PF][A_PASSES], rw[PF][W_PASSES];  // PF register sets: global loads run PF slices ahead
  size_t a_off[A_PASSES], w_off[W_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int row = m0 + r + RPP * p;
    row = row < g.M ? row : g.M - 1;
    a_off[p] = (size_t)row * g.lda + c4 * 4;
  }
#pragma unroll
  for (int p = 0; p < W_PASSES; ++p) {
    int n = n0 + r + RPP * p;
    n = n < g.N ? n : g.N - 1;
    w_off[p] = (size_t)n * g.ldw + c4 * 4;
  }
  // Unconditional loads (no divergent branches in the loop): rows are clamped into the matrix and a
  // K tail (K % 32 != 0, only the embedding MLP's K = 100) reads a clamped in-range address and is
  // zeroed by a select.
  auto load_into = [&](f32x4* xa, f32x4* xw, int k0) {
    const int kcol = k0 + c4 * 4;
    const bool kin = kcol < g.K;
    const int kc = kin ? k0 : (g.K - 4 - c4 * 4);
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      f32x4 v = *reinterpret_cast<const f32x4*>(Asrc + a_off[p] + kc);
      xa[p] = kin ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) {
      f32x4 v = *reinterpret_cast<const f32x4*>(W + w_off[p] + kc);
      xw[p] = kin ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf) {
    float* As = lds + buf * (BM + BN) * LDS_LD;
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<f32x4*>(As + (r + RPP * p) * LDS_LD + c4 * 4) = xa[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<f32x4*>(Ws + (r + RPP * p) * LDS_LD + c4 * 4) = xw[p];
  };


  // ---- per-wave MFMA tile ------------------------------------------------------------------------
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  auto compute_slice = [&](int buf) {
    const float* As = lds + buf * (BM + BN) * LDS_LD + (wm0 + l32) * LDS_LD + half * (BK / 2);
    const float* Ws = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn0 + l32) * LDS_LD + half * (BK / 2);
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      f32x4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a[mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * LDS_LD + kk * 4);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        b[ni] = *reinterpret_cast<const f32x4*>(Ws + ni * 32 * LDS_LD + kk * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            if (ABL == 2) {
              asm volatile("" ::"v"(a[mi][c]), "v"(b[ni][c]));
            } else {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][c], b[ni][c], acc[mi][ni], 0, 0, 0);
            }
          }
    }
  };

  // Software pipeline: slice 0 goes straight to LDS; register set u then always holds slice
  // (current + u + 1).  One slice of MFMA work (0.4 us alone on a SIMD) does not cover an L2 /
  // Infinity-Cache round trip, so a lone block (small-M launches, tail rounds) is latency bound
  // unless several slices are in flight: PF = 2 or 4 on the path.
  const int nslices = (g.K + BK - 1) / BK;
  load_into(ra[0], rw[0], 0);
  store_from(ra[0], rw[0], 0);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u + 1 < nslices && ABL != 1) load_into(ra[u], rw[u], (u + 1) * BK);
  for (int t = 0; t < nslices; t += (PF > 1 ? PF : 2)) {
#pragma unroll
    for (int u = 0; u < (PF > 1 ? PF : 2); ++u) {
      const int tt = t + u;
      if (tt < nslices) {
        constexpr int dummy = 0;
        (void)dummy;
        const int set = PF > 1 ? u : 0;
        compute_slice(u & 1);
        if (ABL != 1) {
          if (tt + 1 < nslices) store_from(ra[set], rw[set], (u + 1) & 1);
          if (tt + 1 + PF < nslices) load_into(ra[set], rw[set], (tt + 1 + PF) * BK);
        }
        if (ABL != 3) __syncthreads();
      }
    }
  }

  // ---- epilogue: bias, activation, residual, store ------------------------------------------------
  // Every residual load of the tile is issued first (clamped addresses, no branches), then the
  // stores follow without waits in between: `residual` may alias C, and interleaving loads with
  // possibly-aliasing stores makes the compiler serialise them (s_waitcnt vmcnt(0) per element).
  // Interior tiles (the common case: M, N multiples of the tile) store without predication.
  float bv[NI];
  int colc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l32;
    colc[ni] = col < g.N ? col : g.N - 1;
    bv[ni] = g.bias ? g.bias[colc[ni]] : 0.f;
  }
  if (g.res) {
    const float* rp = g.res + bz * g.batch_stride_c;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float rv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
          row = row < g.M ? row : g.M - 1;
          rv[e] = rp[(size_t)row * g.ldr + colc[ni]];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[mi][ni][e] + bv[ni];
          if (g.act == 1) v = fmaxf(v, 0.f);
          acc[mi][ni][e] = v + rv[e];
        }
      }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[mi][ni][e] + bv[ni];
          if (g.act == 1) v = fmaxf(v, 0.f);
          acc[mi][ni][e] = v;
        }
  }
  const bool interior = (m0 + BM <= g.M) && (n0 + BN <= g.N);
  if (interior) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float* cp = Cout + (size_t)(m0 + wm0 + mi * 32 + 4 * half) * g.ldc + (n0 + wn0 + ni * 32 + l32);
#pragma unroll
        for (int e = 0; e < 16; ++e) cp[(size_t)((e & 3) + 8 * (e >> 2)) * g.ldc] = acc[mi][ni][e];
      }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = n0 + wn0 + ni * 32 + l32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
          if (row < g.M && col < g.N) Cout[(size_t)row * g.ldc + col] = acc[mi][ni][e];
        }
      }
  }
}

template <int BM, int BN, int WM, int WN, int BK, int ABL = 0, int PF = 1>
int launch_gemm(GemmArgs g, int batch, hipStream_t st) {
  static bool attr_set = false;
  constexpr int bytes = gemm_lds_bytes<BM, BN, BK>();
  if (!attr_set) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, WM, WN, BK, ABL, PF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  g.tiles_m = ff_cdiv(g.M, BM);
  g.tiles_n = ff_cdiv(g.N, BN);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, ABL, PF>), dim3(g.tiles_m * g.tiles_n, batch),
                     dim3(256), bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
// ---- pipelined variant: fragment prefetch + 3-buffer LDS ring, branch-free steady state ---------
// Same tiling as gemm_f32_kernel, different schedule (requires K % BK == 0).  In iteration t a wave
//   * reads the MFMA fragments of slice t+1 from LDS buffer (t+1)%3 into the alternate register set,
//   * writes slice t+2 (already in staging registers) into LDS buffer (t+2)%3,
//   * re-issues the global loads of slice t+4 into the same staging registers,
//   * runs the MFMA chain of slice t on fragments that were read one iteration earlier,
//   * barrier.
// A wave issues in order, so everything that is not an MFMA is placed BETWEEN the MFMAs of the chain
// (sched_group_barrier: one LDS/VMEM instruction in the shadow of each 64-cycle MFMA); nothing but
// the barrier separates the MFMA chains of consecutive slices.  Slices past the end are clamped to
// the last slice (a few redundant loads instead of branches in the loop body).
template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(256) void gemm_pipe_kernel(GemmArgs g) {
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int LDS_LD = BK + 4;
  constexpr int TPR = BK / 4, RPP = 256 / TPR;
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;
  constexpr int KF = BK / 8;  // float4 fragments per operand tile per slice
  constexpr int BUF_FLOATS = (BM + BN) * LDS_LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int nblocks = g.tiles_m * g.tiles_n;
  const int lid = ff_xcd_remap(blockIdx.x, nblocks);
  const int m0 = (lid / g.tiles_n) * BM;
  const int n0 = (lid % g.tiles_n) * BN;
  const long long bz = blockIdx.y;
  const float* __restrict__ Asrc =
      ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  const float* __restrict__ W = g.W + bz * g.batch_stride_w;
  float* __restrict__ Cout = g.C + bz * g.batch_stride_c;

  const int c4 = tid % TPR, r = tid / TPR;
  f32x4 ra[2][A_PASSES], rw[2][W_PASSES];
  const float* a_ptr[A_PASSES];
  const float* w_ptr[W_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int row = m0 + r + RPP * p;
    row = row < g.M ? row : g.M - 1;
    a_ptr[p] = Asrc + (size_t)row * g.lda + c4 * 4;
  }
#pragma unroll
  for (int p = 0; p < W_PASSES; ++p) {
    int n = n0 + r + RPP * p;
    n = n < g.N ? n : g.N - 1;
    w_ptr[p] = W + (size_t)n * g.ldw + c4 * 4;
  }
  const int nsl = g.K / BK;
  auto load_into = [&](f32x4* xa, f32x4* xw, int slice) {
    const int k0 = (slice < nsl ? slice : nsl - 1) * BK;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) xa[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + k0);
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) xw[p] = *reinterpret_cast<const f32x4*>(w_ptr[p] + k0);
  };
  float* const st_a = lds + r * LDS_LD + c4 * 4;
  float* const st_w = st_a + BM * LDS_LD;
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf) {
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<f32x4*>(st_a + buf * BUF_FLOATS + RPP * p * LDS_LD) = xa[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<f32x4*>(st_w + buf * BUF_FLOATS + RPP * p * LDS_LD) = xw[p];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
  const float* const fr_a = lds + (wm0 + l32) * LDS_LD + half * (BK / 2);
  const float* const fr_w = lds + BM * LDS_LD + (wn0 + l32) * LDS_LD + half * (BK / 2);
  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  f32x4 fa[2][MI][KF], fb[2][NI][KF];
  auto read_frags = [&](f32x4 (*xa)[KF], f32x4 (*xb)[KF], int buf) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int kk = 0; kk < KF; ++kk)
        xa[mi][kk] = *reinterpret_cast<const f32x4*>(fr_a + buf * BUF_FLOATS + mi * 32 * LDS_LD + kk * 4);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int kk = 0; kk < KF; ++kk)
        xb[ni][kk] = *reinterpret_cast<const f32x4*>(fr_w + buf * BUF_FLOATS + ni * 32 * LDS_LD + kk * 4);
  };
  auto mfma_frags = [&](f32x4 (*xa)[KF], f32x4 (*xb)[KF]) {
#pragma unroll
    for (int kk = 0; kk < KF; ++kk)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[mi][kk][c], xb[ni][kk][c], acc[mi][ni], 0, 0, 0);
  };
  auto interleave = [&]() {
    constexpr int N_MFMA = MI * NI * KF * 4;
    constexpr int N_DSR = (MI + NI) * KF, N_DSW = A_PASSES + W_PASSES, N_VM = A_PASSES + W_PASSES;
    constexpr int N_OTHER = N_DSR + N_DSW + N_VM;
    constexpr int PER = N_MFMA / N_OTHER > 0 ? N_MFMA / N_OTHER : 1;  // MFMAs per interleaved op
#pragma unroll
    for (int q = 0; q < N_DSR; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < N_DSW; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < N_VM; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };

  // ---- epilogue operands first: bias and residual tile are fetched BEFORE the K loop (their
  //      latency hides under it; the residual tile is only overwritten by this block, at the end) ----
  float bv[NI];
  int colc[NI];
  float rv[MI][NI][16];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l32;
    colc[ni] = col < g.N ? col : g.N - 1;
    bv[ni] = g.bias ? g.bias[colc[ni]] : 0.f;
  }
  if (g.res) {
    const float* rp = g.res + bz * g.batch_stride_c;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
          row = row < g.M ? row : g.M - 1;
          rv[mi][ni][e] = rp[(size_t)row * g.ldr + colc[ni]];
        }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[mi][ni][e] = 0.f;
  }

  // ---- prologue ----
  load_into(ra[0], rw[0], 0);
  load_into(ra[1], rw[1], 1);
  store_from(ra[0], rw[0], 0);
  store_from(ra[1], rw[1], 1);
  load_into(ra[0], rw[0], 2);
  load_into(ra[1], rw[1], 3);
  __syncthreads();
  read_frags(fa[0], fb[0], 0);

  // ---- steady state: two slices per trip (static register-set indices) ----
  int b1 = 1, b2 = 2, b0 = 0;  // LDS buffers of slices t+1, t+2, t
  for (int t = 0; t < nsl; t += 2) {
    // slice t
    read_frags(fa[1], fb[1], b1);
    store_from(ra[0], rw[0], b2);
    load_into(ra[0], rw[0], t + 4);
    mfma_frags(fa[0], fb[0]);
    interleave();
    __syncthreads();
    { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
    if (t + 1 >= nsl) break;  // odd slice count (block-uniform, not taken on the path)
    // slice t + 1
    read_frags(fa[0], fb[0], b1);
    store_from(ra[1], rw[1], b2);
    load_into(ra[1], rw[1], t + 5);
    mfma_frags(fa[1], fb[1]);
    interleave();
    __syncthreads();
    { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
  }

  // ---- epilogue: bias, activation, residual (already in registers), store ----
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float v = acc[mi][ni][e] + bv[ni];
        if (g.act == 1) v = fmaxf(v, 0.f);
        acc[mi][ni][e] = v + rv[mi][ni][e];
      }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wn0 + ni * 32 + l32;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
        if (row < g.M && col < g.N) Cout[(size_t)row * g.ldc + col] = acc[mi][ni][e];
      }
    }
}

template <int BM, int BN, int WM, int WN, int BK>
int launch_pipe(GemmArgs g, int batch, hipStream_t st) {
  if (g.K % BK != 0) return launch_gemm<BM, BN, WM, WN, 32, 0, 2>(g, batch, st);  // K tail: generic kernel
  static bool attr_set = false;
  constexpr int bytes = 3 * (BM + BN) * (BK + 4) * (int)sizeof(float);
  if (!attr_set) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<BM, BN, WM, WN, BK>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  g.tiles_m = ff_cdiv(g.M, BM);
  g.tiles_n = ff_cdiv(g.N, BN);
  hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, WM, WN, BK>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256),
                     bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- persistent variant of the pipelined 64x64 kernel --------------------------------------------
// A fixed grid of blocks (2 per CU) walks the tile list; the software pipeline of gemm_pipe_kernel
// (3-buffer LDS ring, fragment prefetch, staging registers 2 slices ahead, MFMA-interleaved issue) runs
// over the FLAT sequence of (tile, K-slice) pairs, so the global loads of the next tile's first slices
// are already in flight while the current tile finishes: no pipeline fill/drain (an HBM round trip plus
// the store tail, ~9k cycles) per tile, only per block.  Bias and residual of a tile are fetched when
// its first slice is computed and consumed after its last one.  Requires K % 32 == 0 and K >= 128.
__global__ __launch_bounds__(256) void gemm_persist_kernel(GemmArgs g, int total_tiles) {
  constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 4, RPP = 32, KF = BK / 8;
  constexpr int BUF_FLOATS = (BM + BN) * LDS_LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int c4 = tid & 7, r = tid >> 3;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // ---- this block's tile list: XCD x = blockIdx % 8 owns a contiguous range of logical tiles ----
  const int G = gridDim.x;
  int first, stride, limit;
  if ((G & 7) == 0) {
    const int x = blockIdx.x & 7, q = total_tiles >> 3, rem = total_tiles & 7;
    const int lo = (x < rem) ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    limit = lo + q + (x < rem ? 1 : 0);
    first = lo + (blockIdx.x >> 3);
    stride = G >> 3;
  } else {
    first = blockIdx.x; stride = G; limit = total_tiles;
  }
  if (first >= limit) return;
  const int my_tiles = (limit - first + stride - 1) / stride;

  // ---- load cursor (runs 4 slices ahead of the MFMA chain, crosses tile boundaries early) ----
  const float* a_ptr[2];
  const float* w_ptr[2];
  auto set_load_tile = [&](int k) {
    const int id = first + (k < my_tiles ? k : my_tiles - 1) * stride;
    const int bz = id / tiles_mn, rem2 = id - bz * tiles_mn;
    const int m0 = (rem2 / g.tiles_n) * BM, n0 = (rem2 % g.tiles_n) * BN;
    const float* Asrc = ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + (long long)bz * g.batch_stride_a;
    const float* W = g.W + (long long)bz * g.batch_stride_w;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int row = m0 + r + RPP * p;
      row = row < g.M ? row : g.M - 1;
      a_ptr[p] = Asrc + (size_t)row * g.lda + c4 * 4;
      int n = n0 + r + RPP * p;
      n = n < g.N ? n : g.N - 1;
      w_ptr[p] = W + (size_t)n * g.ldw + c4 * 4;
    }
  };
  int ld_k = 0, ld_j = 0;
  set_load_tile(0);
  f32x4 ra[2][2], rw[2][2];
  auto load_next = [&](f32x4* xa, f32x4* xw) {  // loads slice (ld_k, ld_j); branch-free
    const int k0 = ld_j * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xa[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + k0);
      xw[p] = *reinterpret_cast<const f32x4*>(w_ptr[p] + k0);
    }
  };
  auto advance = [&]() {  // block-uniform; past the last tile the cursor stays on the last slice
    if (++ld_j == nsl) {
      if (ld_k + 1 < my_tiles) { ld_j = 0; ++ld_k; set_load_tile(ld_k); }
      else ld_j = nsl - 1;
    }
  };
  float* const st_a = lds + r * LDS_LD + c4 * 4;
  float* const st_w = st_a + BM * LDS_LD;
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(st_a + buf * BUF_FLOATS + RPP * p * LDS_LD) = xa[p];
      *reinterpret_cast<f32x4*>(st_w + buf * BUF_FLOATS + RPP * p * LDS_LD) = xw[p];
    }
  };
  const float* const fr_a = lds + (wm0 + l32) * LDS_LD + half * (BK / 2);
  const float* const fr_w = lds + BM * LDS_LD + (wn0 + l32) * LDS_LD + half * (BK / 2);
  f32x4 fa[2][KF], fb[2][KF];
  auto read_frags = [&](f32x4* xa, f32x4* xb, int buf) {
#pragma unroll
    for (int kk = 0; kk < KF; ++kk) {
      xa[kk] = *reinterpret_cast<const f32x4*>(fr_a + buf * BUF_FLOATS + kk * 4);
      xb[kk] = *reinterpret_cast<const f32x4*>(fr_w + buf * BUF_FLOATS + kk * 4);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto mfma_frags = [&](const f32x4* xa, const f32x4* xb) {
#pragma unroll
    for (int kk = 0; kk < KF; ++kk)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[kk][c], xb[kk][c], acc, 0, 0, 0);
  };
  auto interleave = [&]() {  // 16 MFMAs: 8 ds_read, 4 ds_write, 4 global_load in their shadows
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };

  // ---- compute-side tile state (epilogue operands) ----
  int cp_k = 0, cp_j = 0;
  int e_row0 = 0, e_col = 0;
  long long e_coff = 0;
  bool e_colok = false;
  float bv = 0.f, rv[16];
  auto begin_tile = [&](int k) {  // decode the tile and fetch bias / residual (used nsl slices later)
    const int id = first + k * stride;
    const int bz = id / tiles_mn, rem2 = id - bz * tiles_mn;
    const int m0 = (rem2 / g.tiles_n) * BM, n0 = (rem2 % g.tiles_n) * BN;
    e_row0 = m0 + wm0 + 4 * half;
    e_col = n0 + wn0 + l32;
    e_colok = e_col < g.N;
    e_coff = (long long)bz * g.batch_stride_c;
    const int colc = e_colok ? e_col : g.N - 1;
    bv = g.bias ? g.bias[colc] : 0.f;
    if (g.res) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = e_row0 + (e & 3) + 8 * (e >> 2);
        row = row < g.M ? row : g.M - 1;
        rv[e] = g.res[e_coff + (size_t)row * g.ldr + colc];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = 0.f;
    }
  };
  auto end_tile = [&]() {
    float* cp = g.C + e_coff;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = e_row0 + (e & 3) + 8 * (e >> 2);
      float v = acc[e] + bv;
      if (g.act == 1) v = fmaxf(v, 0.f);
      v += rv[e];
      if (row < g.M && e_colok) cp[(size_t)row * g.ldc + e_col] = v;
      acc[e] = 0.f;
    }
  };

  // ---- prologue: slices 0,1 -> LDS; slices 2,3 -> staging registers ----
  load_next(ra[0], rw[0]); advance();
  load_next(ra[1], rw[1]); advance();
  store_from(ra[0], rw[0], 0);
  store_from(ra[1], rw[1], 1);
  load_next(ra[0], rw[0]); advance();
  load_next(ra[1], rw[1]); advance();
  begin_tile(0);
  __syncthreads();
  read_frags(fa[0], fb[0], 0);

  int b0 = 0, b1 = 1, b2 = 2;
  const int total_slices = my_tiles * nsl;   // nsl is even (checked on the host)
  for (int s = 0; s < total_slices; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      read_frags(fa[u ^ 1], fb[u ^ 1], b1);
      store_from(ra[u], rw[u], b2);
      load_next(ra[u], rw[u]);
      mfma_frags(fa[u], fb[u]);
      interleave();
      advance();
      if (++cp_j == nsl) {  // block-uniform: last slice of the tile just issued
        end_tile();
        cp_j = 0;
        if (++cp_k < my_tiles) begin_tile(cp_k);
      }
      __syncthreads();
      { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
    }
  }
}

int launch_persist(GemmArgs g, int batch, hipStream_t st) {
  static bool attr_set = false;
  constexpr int bytes = 3 * 128 * 36 * (int)sizeof(float);
  if (!attr_set) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_persist_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  g.tiles_m = ff_cdiv(g.M, 64);
  g.tiles_n = ff_cdiv(g.N, 64);
  const long total = (long)g.tiles_m * g.tiles_n * batch;
  int grid = total < 512 ? (int)total : 512;   // 256 CUs x 2 resident blocks (55 KB LDS each)
  hipLaunchKernelGGL(gemm_persist_kernel, dim3(grid), dim3(256), bytes, st, g, (int)total);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- 8-wave, in-block split-K variant of the 64x64 tile ---------------------------------------
// Same 64x64 output tile and LDS image as the BK=64 kernel, but 512 threads: waves 0-3 multiply the
// first 32 k of every 64-wide slice, waves 4-7 the second 32 (both groups cover the whole tile), and
// the two partial accumulators are added through LDS before the epilogue.  One block therefore puts
// TWO MFMA-issuing waves on every SIMD: a lone block on a CU (small-M launches, and the tail round of
// large ones) hides its own LDS/barrier latencies instead of idling the matrix pipe.
template <int PF>
__global__ __launch_bounds__(512) void gemm_ks_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 64, LDS_LD = BK + 4;
  constexpr int TPR = BK / 4, RPP = 512 / TPR;       // 16 threads per row, 32 rows per pass
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;  // 2, 2
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int nblocks = g.tiles_m * g.tiles_n;
  const int lid = ff_xcd_remap(blockIdx.x, nblocks);
  const int m0 = (lid / g.tiles_n) * BM;
  const int n0 = (lid % g.tiles_n) * BN;
  const long long bz = blockIdx.y;
  const float* __restrict__ Asrc =
      ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  const float* __restrict__ W = g.W + bz * g.batch_stride_w;
  float* __restrict__ Cout = g.C + bz * g.batch_stride_c;

  const int c4 = tid % TPR, r = tid / TPR;
  f32x4 ra[A_PASSES], rw[W_PASSES], ra2[A_PASSES], rw2[W_PASSES];
  size_t a_off[A_PASSES], w_off[W_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int row = m0 + r + RPP * p;
    row = row < g.M ? row : g.M - 1;
    a_off[p] = (size_t)row * g.lda + c4 * 4;
  }
#pragma unroll
  for (int p = 0; p < W_PASSES; ++p) {
    int n = n0 + r + RPP * p;
    n = n < g.N ? n : g.N - 1;
    w_off[p] = (size_t)n * g.ldw + c4 * 4;
  }
  auto load_into = [&](f32x4* xa, f32x4* xw, int k0) {
    const bool kin = (k0 + c4 * 4) < g.K;
    const int kc = kin ? k0 : (g.K - 4 - c4 * 4);
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      f32x4 v = *reinterpret_cast<const f32x4*>(Asrc + a_off[p] + kc);
      xa[p] = kin ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) {
      f32x4 v = *reinterpret_cast<const f32x4*>(W + w_off[p] + kc);
      xw[p] = kin ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf) {
    float* As = lds + buf * (BM + BN) * LDS_LD;
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<f32x4*>(As + (r + RPP * p) * LDS_LD + c4 * 4) = xa[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<f32x4*>(Ws + (r + RPP * p) * LDS_LD + c4 * 4) = xw[p];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int kgrp = wave >> 2;                       // which 32-wide half of the 64-wide slice
  const int wm0 = ((wave & 3) >> 1) * 32, wn0 = (wave & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  auto compute_slice = [&](int buf) {
    const float* As = lds + buf * (BM + BN) * LDS_LD + (wm0 + l32) * LDS_LD + kgrp * 32 + half * 16;
    const float* Ws = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn0 + l32) * LDS_LD + kgrp * 32 + half * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a = *reinterpret_cast<const f32x4*>(As + kk * 4);
      f32x4 b = *reinterpret_cast<const f32x4*>(Ws + kk * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc, 0, 0, 0);
    }
  };

  const int nslices = (g.K + BK - 1) / BK;
  load_into(ra, rw, 0);
  store_from(ra, rw, 0);
  __syncthreads();
  if (PF == 2) {
    if (1 < nslices) load_into(ra, rw, 1 * BK);
    if (2 < nslices) load_into(ra2, rw2, 2 * BK);
    for (int t = 0; t < nslices; t += 2) {
      compute_slice(0);
      if (t + 1 < nslices) store_from(ra, rw, 1);
      if (t + 3 < nslices) load_into(ra, rw, (t + 3) * BK);
      __syncthreads();
      if (t + 1 < nslices) {
        compute_slice(1);
        if (t + 2 < nslices) store_from(ra2, rw2, 0);
        if (t + 4 < nslices) load_into(ra2, rw2, (t + 4) * BK);
        __syncthreads();
      }
    }
  } else {
    for (int t = 0; t < nslices; ++t) {
      const bool more = (t + 1) < nslices;
      if (more) load_into(ra, rw, (t + 1) * BK);
      compute_slice(t & 1);
      if (more) store_from(ra, rw, (t + 1) & 1);
      __syncthreads();
    }
  }
  // ---- add the two k-groups through LDS (the loop's final barrier has passed: LDS is free) ----
  float* red = lds + (wave & 3) * (16 * 64);
  if (kgrp == 1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) red[e * 64 + lane] = acc[e];
  }
  __syncthreads();
  if (kgrp == 1) return;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += red[e * 64 + lane];

  const int col = n0 + wn0 + l32;
  const int colc = col < g.N ? col : g.N - 1;
  const float bv = g.bias ? g.bias[colc] : 0.f;
  float rv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    int row = m0 + wm0 + 4 * half + (e & 3) + 8 * (e >> 2);
    row = row < g.M ? row : g.M - 1;
    rv[e] = g.res ? g.res[bz * g.batch_stride_c + (size_t)row * g.ldr + colc] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m0 + wm0 + 4 * half + (e & 3) + 8 * (e >> 2);
    float v = acc[e] + bv;
    if (g.act == 1) v = fmaxf(v, 0.f);
    v += rv[e];
    if (row < g.M && col < g.N) Cout[(size_t)row * g.ldc + col] = v;
  }
}

template <int PF>
int launch_ks(GemmArgs g, int batch, hipStream_t st) {
  static bool attr_set = false;
  constexpr int bytes = 2 * 128 * 68 * (int)sizeof(float);
  if (!attr_set) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ks_kernel<PF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  g.tiles_m = ff_cdiv(g.M, 64);
  g.tiles_n = ff_cdiv(g.N, 64);
  hipLaunchKernelGGL((gemm_ks_kernel<PF>), dim3(g.tiles_m * g.tiles_n, batch), dim3(512), bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

}  // namespace

extern "C" int ff_gemm_f32_batched(const float* A, int lda, const float* A2, int n_split,
                                   const float* W, int ldw, const float* bias, const float* residual,
                                   int ldr, float* C, int ldc, int M, int N, int K, int act, int tile,
                                   int batch, long long stride_a, long long stride_w,
                                   long long stride_c, ff_stream_t stream) {
  if (M == 0 || N == 0 || batch == 0) return FF_OK;
  FF_CHECK_ARG(M > 0 && N > 0 && K > 0 && (K & 3) == 0, "ff_gemm_f32: bad M=%d N=%d K=%d (K %% 4)", M, N, K);
  FF_CHECK_ARG(A && W && C, "ff_gemm_f32: null operand");
  FF_CHECK_ARG((lda & 3) == 0 && (ldw & 3) == 0 && lda >= K && ldw >= K && ldc >= N,
               "ff_gemm_f32: bad leading dimensions lda=%d ldw=%d ldc=%d", lda, ldw, ldc);
  FF_CHECK_ARG(ff_aligned16(A) && ff_aligned16(W) && (!A2 || ff_aligned16(A2)),
               "ff_gemm_f32: A/A2/W must be 16-byte aligned");
  FF_CHECK_ARG(!residual || ldr >= N, "ff_gemm_f32: bad ldr");
  FF_CHECK_ARG(act == 0 || act == 1, "ff_gemm_f32: act must be 0 or 1");
  FF_CHECK_ARG(tile >= 0 && tile <= 21, "ff_gemm_f32: tile must be 0..6, 13..21 (7..12: timing ablations)");
  FF_CHECK_ARG(batch > 0 && batch <= 65535 && (stride_a & 3) == 0 && (stride_w & 3) == 0,
               "ff_gemm_f32: bad batch arguments");
  FF_CHECK_ARG(batch == 1 || !residual, "ff_gemm_f32: residual is not supported with batch > 1");
  if (A2) FF_CHECK_ARG(n_split > 0 && n_split < N && (n_split % 64) == 0, "ff_gemm_f32: n_split must be a multiple of 64 inside (0,N)");
  GemmArgs g{A, A2, W, bias, residual, C, lda, ldw, ldr, ldc, M, N, K, A2 ? n_split : N, act, 0, 0,
             stride_a, stride_w, stride_c};
  const bool split128 = !A2 || (n_split % 128) == 0;
  if (tile == 0) {
    // Measured on MI355X over the path's shapes (tools/gemm_probe_multi.py under rocprofv3, M = 256..9216,
    // K = 512/1024): the pipelined 64x64 tile wins or ties everywhere -- it is the only shape that
    // keeps several blocks resident per CU at these sizes; its persistent form (21) adds 4-6 % by
    // carrying the software pipeline across tile boundaries.
    tile = 21;
  }
  if (tile == 3 && !split128) tile = 2;
  hipStream_t st = (hipStream_t)stream;
  FFProfScope prof(FF_CAT_GEMM, 2.0 * M * N * K * batch, st);
  switch (tile) {
    case 1: return launch_gemm<64, 64, 32, 32, 32>(g, batch, st);
    case 2: return launch_gemm<128, 64, 64, 32, 32>(g, batch, st);
    case 3: return launch_gemm<128, 128, 64, 64, 32>(g, batch, st);
    case 4: return launch_gemm<64, 64, 32, 32, 64>(g, batch, st);
    case 5: return launch_gemm<64, 64, 32, 32, 32, 0, 2>(g, batch, st);
    case 6: return launch_gemm<64, 64, 32, 32, 32, 0, 4>(g, batch, st);
    case 15: return launch_gemm<64, 64, 32, 32, 64, 0, 2>(g, batch, st);
    case 16: return launch_gemm<64, 64, 32, 32, 32, 0, 8>(g, batch, st);
    case 17: return launch_pipe<64, 64, 32, 32, 32>(g, batch, st);
    case 18: return launch_pipe<64, 64, 32, 32, 64>(g, batch, st);
    case 19: return launch_pipe<128, 128, 64, 64, 32>(g, batch, st);
    case 20: return launch_pipe<128, 64, 64, 32, 32>(g, batch, st);
    case 21:
      if (K % 64 == 0 && K >= 128) return launch_persist(g, batch, st);
      return launch_pipe<64, 64, 32, 32, 32>(g, batch, st);
    case 13: return launch_ks<1>(g, batch, st);
    case 14: return launch_ks<2>(g, batch, st);
    case 7: return launch_gemm<64, 64, 32, 32, 32, 1>(g, batch, st);
    case 8: return launch_gemm<64, 64, 32, 32, 32, 2>(g, batch, st);
    case 9: return launch_gemm<64, 64, 32, 32, 32, 3>(g, batch, st);
    case 10: return launch_gemm<128, 128, 64, 64, 32, 1>(g, batch, st);
    case 11: return launch_gemm<128, 128, 64, 64, 32, 2>(g, batch, st);
    default: return launch_gemm<128, 128, 64, 64, 32, 3>(g, batch, st);
  }
}

extern "C" int ff_gemm_f32(const float* A, int lda, const float* A2, int n_split, const float* W,
                           int ldw, const float* bias, const float* residual, int ldr, float* C,
                           int ldc, int M, int N, int K, int act, int tile, ff_stream_t stream) {
  return ff_gemm_f32_batched(A, lda, A2, n_split, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, tile,
                             1, 0, 0, 0, stream);
}
