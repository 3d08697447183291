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

constexpr int BK = 32;
constexpr int LDS_LD = 36;

template <int BM, int BN>
constexpr int gemm_lds_bytes() { return 2 * (BM + BN) * LDS_LD * (int)sizeof(float); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / 32, W_PASSES = BN / 32;
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
  const int c4 = tid & 7, r = tid >> 3;
  f32x4 ra[A_PASSES], rw[W_PASSES];
  size_t a_off[A_PASSES], w_off[W_PASSES];
#pragma unroll
  for (int p = 0; p < A_PASSES; ++p) {
    int row = m0 + r + 32 * p;
    row = row < g.M ? row : g.M - 1;
    a_off[p] = (size_t)row * g.lda + c4 * 4;
  }
#pragma unroll
  for (int p = 0; p < W_PASSES; ++p) {
    int n = n0 + r + 32 * p;
    n = n < g.N ? n : g.N - 1;
    w_off[p] = (size_t)n * g.ldw + c4 * 4;
  }
  auto load_slice = [&](int k0) {
    const bool kin = (k0 + c4 * 4) < g.K;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      ra[p] = kin ? *reinterpret_cast<const f32x4*>(Asrc + a_off[p] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      rw[p] = kin ? *reinterpret_cast<const f32x4*>(W + w_off[p] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto store_slice = [&](int buf) {
    float* As = lds + buf * (BM + BN) * LDS_LD;
    float* Ws = As + BM * LDS_LD;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p)
      *reinterpret_cast<f32x4*>(As + (r + 32 * p) * LDS_LD + c4 * 4) = ra[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<f32x4*>(Ws + (r + 32 * p) * LDS_LD + c4 * 4) = rw[p];
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
    const float* As = lds + buf * (BM + BN) * LDS_LD + (wm0 + l32) * LDS_LD + half * 16;
    const float* Ws = lds + buf * (BM + BN) * LDS_LD + BM * LDS_LD + (wn0 + l32) * LDS_LD + half * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
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
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][c], b[ni][c], acc[mi][ni], 0, 0, 0);
    }
  };

  const int nslices = (g.K + BK - 1) / BK;
  load_slice(0);
  store_slice(0);
  __syncthreads();
  for (int t = 0; t < nslices; ++t) {
    const bool more = (t + 1) < nslices;
    if (more) load_slice((t + 1) * BK);
    compute_slice(t & 1);
    if (more) store_slice((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: bias, activation, residual, store ------------------------------------------------
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l32;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        if (row < g.M) {
          float v = acc[mi][ni][e] + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          if (g.res) v += g.res[(size_t)row * g.ldr + col];
          Cout[(size_t)row * g.ldc + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_gemm(GemmArgs g, int batch, hipStream_t st) {
  static bool attr_set = false;
  constexpr int bytes = gemm_lds_bytes<BM, BN>();
  if (!attr_set) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, WM, WN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    attr_set = true;
  }
  g.tiles_m = ff_cdiv(g.M, BM);
  g.tiles_n = ff_cdiv(g.N, BN);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256),
                     bytes, st, g);
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
  FF_CHECK_ARG(tile >= 0 && tile <= 3, "ff_gemm_f32: tile must be 0..3");
  FF_CHECK_ARG(batch > 0 && batch <= 65535 && (stride_a & 3) == 0 && (stride_w & 3) == 0,
               "ff_gemm_f32: bad batch arguments");
  FF_CHECK_ARG(batch == 1 || !residual, "ff_gemm_f32: residual is not supported with batch > 1");
  if (A2) FF_CHECK_ARG(n_split > 0 && n_split < N && (n_split % 64) == 0, "ff_gemm_f32: n_split must be a multiple of 64 inside (0,N)");
  GemmArgs g{A, A2, W, bias, residual, C, lda, ldw, ldr, ldc, M, N, K, A2 ? n_split : N, act, 0, 0,
             stride_a, stride_w, stride_c};
  const bool split128 = !A2 || (n_split % 128) == 0;
  if (tile == 0) {
    // Measured on MI355X over the path's shapes (tools/bench_gemm.py, M = 256..9216, K = 512/1024):
    // the 64x64 block tile wins or ties everywhere below ~3000 tiles because it is the only one
    // that keeps several blocks resident per CU; larger tiles only pay for very large M.
    const long t128 = (long)ff_cdiv(M, 128) * ff_cdiv(N, 128) * batch;
    tile = (split128 && t128 >= 4096) ? 3 : 1;
  }
  if (tile == 3 && !split128) tile = 2;
  hipStream_t st = (hipStream_t)stream;
  FFProfScope prof(FF_CAT_GEMM, 2.0 * M * N * K * batch, st);
  switch (tile) {
    case 1: return launch_gemm<64, 64, 32, 32>(g, batch, st);
    case 2: return launch_gemm<128, 64, 64, 32>(g, batch, st);
    default: return launch_gemm<128, 128, 64, 64>(g, batch, st);
  }
}

extern "C" int ff_gemm_f32(const float* A, int lda, const float* A2, int n_split, const float* W,
                           int ldw, const float* bias, const float* residual, int ldr, float* C,
                           int ldc, int M, int N, int K, int act, int tile, ff_stream_t stream) {
  return ff_gemm_f32_batched(A, lda, A2, n_split, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, tile,
                             1, 0, 0, 0, stream);
}
