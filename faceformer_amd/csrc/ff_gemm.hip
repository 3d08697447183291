// Dense fp32 projection on the CDNA4 matrix cores:  C = act(Asel * W^T + bias) + residual.
//
// v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles per instruction per SIMD, 157.3 TF/s chip peak): lane l
// supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31]; the accumulator lane layout is
// col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5).  Both operands of our product are
// K-contiguous (activations [M,K], nn.Linear weights [N,K]), so a block stages a BM x 32 slice of A and
// a BN x 32 slice of W through LDS (row stride 36 floats: 16-byte aligned and conflict-free for
// ds_read_b128 / ds_write_b128, SQ_LDS_BANK_CONFLICT = 0) and each lane reads its 16 k-values (lane
// half 0: k 0..15, half 1: k 16..31 of the slice -- any k permutation is legal as long as A and W use
// the same one) with four 16-byte LDS reads per 32x32 sub-tile.
//
// Four kernels, one tiling (256 threads = 4 waves, one per SIMD, each owning 32x32 accumulators):
//   gemm_generic_kernel  two LDS buffers, staging registers two slices ahead; any K (zero-filled tail).
//   gemm_pipe_kernel     three-buffer LDS ring + fragment prefetch + MFMA-interleaved issue (below).
//   gemm_persist_kernel  the same pipeline run over the flat (tile, slice) sequence by a fixed grid.
//   gemm_streamk_kernel  the persistent pipeline over equal ranges of K units instead of whole tiles;
//                        tiles cut between blocks are summed by the owning block (deterministic).
//                        The path (K = 512 / 1024) uses the last two, picked per launch by a cost model.
// What the measurements on MI355X said while building them (tools/gemm_probe_multi.py under rocprofv3,
// tools/gemm_pmc.sh): a wave issues in order, so LDS / VMEM instructions placed before the dependent
// MFMA chain delay it while the same instructions placed BETWEEN two MFMAs are free (64-cycle shadow):
// 90 -> 100 TF/s on the large steps and 17 -> 13 us for a lone block; a single slice of MFMA work does
// not cover an L2 / Infinity-Cache round trip (loads must run >= 2 slices ahead); 64x64 block tiles beat
// 128x64 / 128x128 at every M of the path because they are the only shape that keeps >= 2 blocks per
// CU; wave-private tiles (no barrier, 2x the L2 traffic) and in-block split-K lost; barriers and the
// phase of co-resident blocks do not matter; unpadded XOR-swizzled LDS rows (48 KB per block) with THREE
// blocks per CU instead of two: 116.6 vs 115.4 TF/s at 8192x1536x512 -- occupancy is not the limit.
#include <atomic>
#include <mutex>
#include <vector>

#include <stdlib.h>

#include "ff_common.h"
#include "ff_device.h"

// Timing experiment (tools/gemm_slice_probe.py, -DFF_EXP_STAMP): the four waves of workgroup 0 of the persistent kernel stamp
// the shader clock in front of and behind the block barrier of their first 256 slices.
#ifdef FF_EXP_STAMP
__device__ unsigned long long ff_exp_stamps[4][256][2];
#define FF_EXP_STAMP_PRE()                                                                          \
  unsigned long long _t0 = 0;                                                                       \
  if (blockIdx.x == 0 && _slice < 256) _t0 = __builtin_readcyclecounter();
#define FF_EXP_STAMP_POST()                                                                         \
  if (blockIdx.x == 0 && _slice < 256) {                                                            \
    const unsigned long long _t1 = __builtin_readcyclecounter();                                    \
    if (lane == 0) { ff_exp_stamps[wave][_slice][0] = _t0; ff_exp_stamps[wave][_slice][1] = _t1; }  \
  }                                                                                                 \
  ++_slice;
extern "C" int ff_exp_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_exp_stamps), sizeof(ff_exp_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define FF_EXP_STAMP_PRE()
#define FF_EXP_STAMP_POST()
#endif

// Timing experiment (tools/panel_phase_probe.py, -DFF_EXP_PANEL_STAMP): workgroup 0 of gemm_panel_kernel stamps the shader
// clock at entry, when its panels are in LDS, when the MFMA chains are done and after the stores were issued.
#ifdef FF_EXP_PANEL_STAMP
__device__ unsigned long long ff_exp_panel_stamps[8];
extern "C" int ff_exp_read_panel_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_exp_panel_stamps), sizeof(ff_exp_panel_stamps)) == hipSuccess ? 0 : -1;
}
#define FF_EXP_PSTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ff_exp_panel_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FF_EXP_PSTAMP(i) do { } while (0)
#endif

// Timing experiment (tools/streamk_probe.py, -DFF_EXP_SK_STAMP): block lb == 100 of gemm_streamk_kernel stamps the shader clock at
// entry, around its hand-over (contributed partial tile) and around its fix-up (owned tile) and at its end.
#ifdef FF_EXP_SK_STAMP
__device__ unsigned long long ff_exp_sk_stamps[8];
extern "C" int ff_exp_read_sk_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_exp_sk_stamps), sizeof(ff_exp_sk_stamps)) == hipSuccess ? 0 : -1;
}
#define FF_EXP_SKSTAMP(i) do { if (lb == 100 && threadIdx.x == 0) ff_exp_sk_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FF_EXP_SKSTAMP(i) do { } while (0)
#endif

namespace {


// Row statistics of a LayerNorm input from the per-32-column (mean, M2) partials its producer left behind:
// Chan's parallel update for equal-sized parts (no E[x^2] - mean^2 cancellation).  -> (mean, 1/sqrt(var + eps)).
// (The small-M kernel merges a block's 32 rows once, eight lanes per row, and shares them through LDS.)
//
// Staging side of the persistent kernels: the 8 lanes that stage one A row (lane & 7 = 16-byte column of the
// slice) share the work of merging its segment statistics: lane c takes segments c, c + 8, ... (K <= 1024:
// at most 4), the partial sums meet through DPP moves inside the group of 8 lanes.  The raw loads are issued
// when the load cursor enters a tile and consumed ~3 slices later (ff_ln_finish), so their latency never sits
// in front of the MFMA chain.
constexpr int LN_NQ = 2;                     // segments per lane: ln_nseg <= 8 * LN_NQ, i.e. K <= 512 for the fused consumer
struct LnRaw { f32x2 seg[LN_NQ]; };          // (mean, M2) pairs exactly as loaded: no register shuffling behind the loads
__device__ __forceinline__ void ff_ln_issue(const float* __restrict__ st, int nseg, int c, LnRaw& r) {
#pragma unroll
  for (int q = 0; q < LN_NQ; ++q) {
    const int sidx = c + 8 * q;
    r.seg[q] = ff_ld8(st + 2 * (sidx < nseg ? sidx : 0));
  }
}
__device__ __forceinline__ void ff_ln_finish(const LnRaw& r, int nseg, int c, float eps, float& mean, float& rstd) {
  float sm = 0.f, m2 = 0.f;
#pragma unroll
  for (int q = 0; q < LN_NQ; ++q)
    if (c + 8 * q < nseg) { sm += r.seg[q].x; m2 += r.seg[q].y; }
  sm = ff_sum8(sm);
  m2 = ff_sum8(m2);
  mean = sm / (float)nseg;
  float dev = 0.f;
#pragma unroll
  for (int q = 0; q < LN_NQ; ++q)
    if (c + 8 * q < nseg) { const float d = r.seg[q].x - mean; dev += d * d; }
  dev = ff_sum8(dev);
  const float var = (m2 + 32.f * dev) / (32.f * (float)nseg);
  rstd = 1.0f / sqrtf(var + eps);
}

// rv[e] = rowtab[(row_e / div), col] for the 16 accumulator rows row_e = row0 + (e&3) + 8*(e>>2) of a lane: one
// division; the 28-row span crosses a multiple of div at most once when div >= 28 (sequences per micro-batch).
__device__ __forceinline__ void ff_load_rowtab(const GemmArgs& g, int row0, int colc, float (&rv)[16]) {
  const int div = g.rowtab_div;
  const int qmax = (g.M - 1) / div;
  const float* tp = g.rowtab + colc;
  if (div >= 28) {
    const int q0 = row0 / div, rem0 = row0 - q0 * div;
    if (rem0 + 27 < div) {   // the lane's 28-row span lies inside ONE table row (always, when div is a multiple of 64): one load
      const float v = ff_ldw(tp + (size_t)(q0 < qmax ? q0 : qmax) * g.ld_rowtab);
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = v;
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int q = q0 + ((rem0 + (e & 3) + 8 * (e >> 2)) >= div ? 1 : 0);
        q = q < qmax ? q : qmax;
        rv[e] = ff_ldw(tp + (size_t)q * g.ld_rowtab);
      }
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      int q = (row0 + (e & 3) + 8 * (e >> 2)) / div;
      q = q < qmax ? q : qmax;
      rv[e] = ff_ldw(tp + (size_t)q * g.ld_rowtab);
    }
  }
}

// Epilogue side: one wave holds a finished 32x32 sub-tile in the MFMA accumulator layout (lane: column l32,
// rows (e&3) + 8*(e>>2) + 4*half).  The values go through a wave-private LDS patch [32][33] so that lane
// (row = l32, half) can sum 16 consecutive columns of ITS row; the two halves meet with one shuffle.  Two passes
// (mean, then centred squares) like the standalone LayerNorm kernel.  Writes (mean, M2) of rows < M.
__device__ __forceinline__ void ff_emit_ln_stats(const float (&v)[16], float* patch, int l32, int half, int row0, int M,
                                                 float* __restrict__ ln_out, int nseg_out, int seg) {
#pragma unroll
  for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + 4 * half) * 33 + l32] = v[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float x[16], sm = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { x[c] = patch[l32 * 33 + half * 16 + c]; sm += x[c]; }
  sm = ff_halves_sum(sm);
  const float mean = sm * (1.0f / 32.0f);
  float m2 = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) { const float d = x[c] - mean; m2 += d * d; }
  m2 = ff_halves_sum(m2);
  const int row = row0 + l32;
  if (half == 0 && row < M) {
    ff_st8(ln_out + ((size_t)row * nseg_out + seg) * 2, f32x2{mean, m2});
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();   // the patch is reused by this wave's next tile
}

// One 32x32 MFMA per k pair: MFMA-interleave hints shared by the pipelined kernels.
template <int N_MFMA, int N_DSR, int N_DSW, int N_VM>
__device__ __forceinline__ void ff_interleave_hints() {
  constexpr int N_OTHER = N_DSR + N_DSW + N_VM;
  constexpr int PER = N_MFMA / N_OTHER > 0 ? N_MFMA / N_OTHER : 1;  // MFMAs per interleaved instruction
#pragma unroll
  for (int q = 0; q < N_DSR; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
  }
#pragma unroll
  for (int q = 0; q < N_DSW; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);    // DS write
  }
#pragma unroll
  for (int q = 0; q < N_VM; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // VMEM read
  }
}

// The persistent kernels' slice: 16 MFMA, 8 fragment reads, 4 LDS writes, 4 global loads; MODE 1 additionally
// normalises the two A float4s it writes (subtract, scale: up to 16 VALU), which belong in the MFMA shadows too.
template <int MODE>
__device__ __forceinline__ void ff_persist_hints() {
  // (MODE 1: the normalisation VALU shares the fragment-read section.  Measured alternatives, by the emitted ISA:
  //  VALU groups behind the LDS-write groups, or only in the second half of the read section, make the scheduler
  //  give up the whole interleave -- 16 back-to-back MFMAs followed by every LDS / VMEM instruction)
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    if (MODE == 1) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // VALU: normalisation of the staged rows
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
  }
}

// ---- generic kernel: any K ------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_generic_kernel(GemmArgs g) {
  constexpr int BK = 32, LDS_LD = BK + 4, PF = 2;
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / 32, W_PASSES = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lid = ff_xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int m0 = (lid / g.tiles_n) * BM;
  const int n0 = (lid % g.tiles_n) * BN;
  const long long bz = blockIdx.y;
  const float* __restrict__ Asrc =
      ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  const float* __restrict__ W = g.W + bz * g.batch_stride_w;
  float* __restrict__ Cout = g.C + bz * g.batch_stride_c;

  // global -> register staging (thread: float4 column c4 of rows r, r+32, ...); rows are clamped into
  // the matrix, a K tail reads a clamped in-range address and is zeroed by a select (no branches).
  const int c4 = tid & 7, r = tid >> 3;
  f32x4 ra[PF][A_PASSES], rw[PF][W_PASSES];
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
      *reinterpret_cast<f32x4*>(As + (r + 32 * p) * LDS_LD + c4 * 4) = xa[p];
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p)
      *reinterpret_cast<f32x4*>(Ws + (r + 32 * p) * LDS_LD + c4 * 4) = xw[p];
  };

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
      for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(As + mi * 32 * LDS_LD + kk * 4);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(Ws + ni * 32 * LDS_LD + kk * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][c], b[ni][c], acc[mi][ni], 0, 0, 0);
    }
  };

  // slice 0 goes straight to LDS; register set u then holds slice (current + u + 1)
  const int nslices = (g.K + BK - 1) / BK;
  load_into(ra[0], rw[0], 0);
  store_from(ra[0], rw[0], 0);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u + 1 < nslices) load_into(ra[u], rw[u], (u + 1) * BK);
  for (int t = 0; t < nslices; t += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int tt = t + u;
      if (tt < nslices) {
        compute_slice(u & 1);
        if (tt + 1 < nslices) store_from(ra[u], rw[u], (u + 1) & 1);
        if (tt + 1 + PF < nslices) load_into(ra[u], rw[u], (tt + 1 + PF) * BK);
        __syncthreads();
      }
    }
  }

  // epilogue: bias, activation, residual, store.  All residual loads of a sub-tile are issued before
  // its stores: `residual` may alias C, and interleaving loads with possibly-aliasing stores makes the
  // compiler serialise them (s_waitcnt vmcnt(0) per element).
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l32;
    const int colc = col < g.N ? col : g.N - 1;
    const float bv = g.bias ? g.bias[colc] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int rbase = m0 + wm0 + mi * 32 + 4 * half;
      float rv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = rbase + (e & 3) + 8 * (e >> 2);
        row = row < g.M ? row : g.M - 1;
        rv[e] = g.res ? g.res[bz * g.batch_stride_c + (size_t)row * g.ldr + colc] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = rbase + (e & 3) + 8 * (e >> 2);
        float v = acc[mi][ni][e] + bv;
        if (g.act == 1) v = fmaxf(v, 0.f);
        v += rv[e];
        if (row < g.M && col < g.N) Cout[(size_t)row * g.ldc + col] = v;
      }
    }
  }
}

// ---- pipelined kernel: fragment prefetch + 3-buffer LDS ring, branch-free steady state (K % 32 == 0) ----
// In iteration t a wave
//   * reads the MFMA fragments of slice t+1 from LDS buffer (t+1)%3 into the alternate register set,
//   * writes slice t+2 (already in staging registers) into LDS buffer (t+2)%3,
//   * re-issues the global loads of slice t+4 into the same staging registers,
//   * runs the MFMA chain of slice t on fragments that were read one iteration earlier,
//   * barrier.
// Everything that is not an MFMA sits between the MFMAs of the chain (sched_group_barrier); nothing but
// the barrier separates the chains of consecutive slices.  Slices past the end are clamped to the last
// slice (a few redundant loads instead of branches in the loop body).
template <int BM, int BN, int WM, int WN, int BK>
__global__ __launch_bounds__(256, (BK == 16 ? 2 : 1)) void gemm_pipe_kernel(GemmArgs g) {
  constexpr int LDS_LD = BK + 4, KF = BK / 8;
  constexpr int TPR = BK / 4, RPP = 256 / TPR;  // threads per staged row, rows per staging pass
  constexpr int MI = WM / 32, NI = WN / 32;
  constexpr int WAVES_N = BN / WN;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
  constexpr int A_PASSES = BM / RPP, W_PASSES = BN / RPP;
  constexpr int BUF_FLOATS = (BM + BN) * LDS_LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lid = ff_xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
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
    ff_interleave_hints<MI * NI * KF * 4, (MI + NI) * KF, A_PASSES + W_PASSES, A_PASSES + W_PASSES>();
  };

  // epilogue operands first: bias and residual tile are fetched BEFORE the K loop (their latency hides
  // under it; the residual tile is only overwritten by this block, at the end)
  float bv[NI];
  int colc[NI];
  constexpr bool PRE = MI * NI < 4;  // four accumulators: no registers left for a residual prefetch
  float rv[PRE ? MI : 1][PRE ? NI : 1][16];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n0 + wn0 + ni * 32 + l32;
    colc[ni] = col < g.N ? col : g.N - 1;
    bv[ni] = g.bias ? g.bias[colc[ni]] : 0.f;
  }
  if (PRE && g.res) {
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
  } else if (PRE) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) rv[mi][ni][e] = 0.f;
  }

  // prologue
  load_into(ra[0], rw[0], 0);
  load_into(ra[1], rw[1], 1);
  store_from(ra[0], rw[0], 0);
  store_from(ra[1], rw[1], 1);
  load_into(ra[0], rw[0], 2);
  load_into(ra[1], rw[1], 3);
  __syncthreads();
  read_frags(fa[0], fb[0], 0);

  // steady state: two slices per trip (static register-set indices)
  int b0 = 0, b1 = 1, b2 = 2;  // LDS buffers of slices t, t+1, t+2
  for (int t = 0; t < nsl; t += 2) {
    read_frags(fa[1], fb[1], b1);
    store_from(ra[0], rw[0], b2);
    load_into(ra[0], rw[0], t + 4);
    mfma_frags(fa[0], fb[0]);
    interleave();
    __syncthreads();
    { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
    if (t + 1 >= nsl) break;  // odd slice count (block-uniform, not taken on the path)
    read_frags(fa[0], fb[0], b1);
    store_from(ra[1], rw[1], b2);
    load_into(ra[1], rw[1], t + 5);
    mfma_frags(fa[1], fb[1]);
    interleave();
    __syncthreads();
    { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
  }

#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = n0 + wn0 + ni * 32 + l32;
      float rl[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (PRE) {
          rl[e] = rv[PRE ? mi : 0][PRE ? ni : 0][e];
        } else {
          int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
          row = row < g.M ? row : g.M - 1;
          rl[e] = g.res ? g.res[bz * g.batch_stride_c + (size_t)row * g.ldr + colc[ni]] : 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm0 + mi * 32 + 4 * half + (e & 3) + 8 * (e >> 2);
        float v = acc[mi][ni][e] + bv[ni];
        if (g.act == 1) v = fmaxf(v, 0.f);
        v += rl[e];
        if (row < g.M && col < g.N) Cout[(size_t)row * g.ldc + col] = v;
      }
    }
}

// ---- persistent form of the pipelined 64x64 kernel (the default on the path) ------------------------------
// A fixed grid of blocks (2 per CU: 55 KB LDS each) walks the tile list; the software pipeline runs over
// the FLAT sequence of (tile, K-slice) pairs, so the global loads of the next tile's first slices are in
// flight while the current tile finishes: no pipeline fill/drain per tile, only per block (+4-6 %).  Bias
// and residual of a tile are fetched when its first slice is computed and consumed after its last one.
// Requires K % 64 == 0 and K >= 128.
template <int MODE>  // MODE: 0 plain, 1 LayerNorm-normalised A rows + row-indexed additive table, 2 emits row statistics of C
__device__ __forceinline__ void gemm_persist_body(const GemmArgs& g, int total_tiles, int vbid, int G, float* lds) {
  constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 4, KF = BK / 8;
  constexpr int BUF_FLOATS = (BM + BN) * LDS_LD;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int c4 = tid & 7, r = tid >> 3;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // this block's tile list: XCD x = (block id) % 8 owns a contiguous range of logical tiles (the column
  // tiles of an A row-panel share an L2); speed only, never correctness
  int first, stride, limit;
  if ((G & 7) == 0) {
    const int x = vbid & 7, q = total_tiles >> 3, rem = total_tiles & 7;
    const int lo = (x < rem) ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    limit = lo + q + (x < rem ? 1 : 0);
    first = lo + (vbid >> 3);
    stride = G >> 3;
  } else {
    first = vbid; stride = G; limit = total_tiles;
  }
  if (first >= limit) return;
  const int my_tiles = (limit - first + stride - 1) / stride;

  // load cursor: runs 4 slices ahead of the MFMA chain and crosses tile boundaries early
  const float* a_ptr[2];
  const float* w_ptr[2];
  // MODE 1: (mean, rstd) of the two rows this thread stages.  Entering a tile issues the raw statistics loads
  // (ln_raw); they replace `cur` exactly when the first slice of that tile is written to LDS: `pend` slices later
  // (the load cursor runs 2 slices ahead of the LDS writes, switches are >= 4 slices apart inside the loop).
  float cur_mu[2] = {0.f, 0.f}, cur_rs[2] = {1.f, 1.f};
  LnRaw ln_raw[2];
  int pend = 0;
  auto ln_finish = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) ff_ln_finish(ln_raw[p], g.ln_nseg, c4, g.ln_eps, cur_mu[p], cur_rs[p]);
  };
  auto set_load_tile = [&](int k, bool defer) {
    const int id = first + (k < my_tiles ? k : my_tiles - 1) * stride;
    const int bz = id / tiles_mn, rem2 = id - bz * tiles_mn;
    const int m0 = (rem2 / g.tiles_n) * BM, n0 = (rem2 % g.tiles_n) * BN;
    const float* Asrc = ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + (long long)bz * g.batch_stride_a;
    const float* W = g.W + (long long)bz * g.batch_stride_w;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int row = m0 + r + 32 * p;
      row = row < g.M ? row : g.M - 1;
      a_ptr[p] = Asrc + (size_t)row * g.lda + c4 * 4;
      if (MODE == 1) ff_ln_issue(g.ln_in + (size_t)row * g.ln_nseg * 2, g.ln_nseg, c4, ln_raw[p]);
      int n = n0 + r + 32 * p;
      n = n < g.N ? n : g.N - 1;
      w_ptr[p] = W + (size_t)n * g.ldw + c4 * 4;
    }
    if (MODE == 1) { if (defer) pend = 3; else ln_finish(); }
  };
  int ld_k = 0, ld_j = 0;
  set_load_tile(0, false);
  f32x4 ra[2][2], rw[2][2];
  auto load_next = [&](f32x4* xa, f32x4* xw, int u) {  // loads slice (ld_k, ld_j); branch-free
    const int k0 = ld_j * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xa[p] = ff_ld16(a_ptr[p] + k0);
      xw[p] = ff_ldw16(w_ptr[p] + k0);
    }
  };
  auto advance = [&](bool defer) {  // block-uniform; past the last tile the cursor stays on the last slice
    if (++ld_j == nsl) {
      if (ld_k + 1 < my_tiles) { ld_j = 0; ++ld_k; set_load_tile(ld_k, defer); }
      else ld_j = nsl - 1;
    }
  };
  float* const st_a = lds + r * LDS_LD + c4 * 4;
  float* const st_w = st_a + BM * LDS_LD;
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf, int u) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(st_a + buf * BUF_FLOATS + 32 * p * LDS_LD) =
          MODE == 1 ? (xa[p] - cur_mu[p]) * cur_rs[p] : xa[p];
      *reinterpret_cast<f32x4*>(st_w + buf * BUF_FLOATS + 32 * p * LDS_LD) = xw[p];
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

  // compute-side tile state (epilogue operands)
  int cp_k = 0;
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
    bv = g.bias ? ff_ldw(g.bias + colc) : 0.f;
    if (g.res) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int row = e_row0 + (e & 3) + 8 * (e >> 2);
        row = row < g.M ? row : g.M - 1;
        rv[e] = ff_ld4(g.res + e_coff + (size_t)row * g.ldr + colc);
      }
    } else if (MODE == 1 && g.rowtab && colc < g.rowtab_cols) {  // additive table indexed by row / div (positions)
      ff_load_rowtab(g, e_row0, colc, rv);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = 0.f;
    }
  };
  auto end_tile = [&]() {
    float* cp = g.C + e_coff;
    float fin[16];
    // All values FIRST, then the stores back to back.  (With the store inside the loop each of its 16 guarded blocks began
    // with `s_waitcnt vmcnt(0)` -- the compiler re-establishes "bias / residual have arrived" in every block, and on gfx9 that
    // counter also counts the store issued by the block before: sixteen serialised write round trips, ~600 cycles each,
    // 9.8 k cycles per tile against 2.0 k per K-slice: tools/gemm_slice_probe.py.)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool tab = MODE == 1 && g.rowtab != nullptr;   // a position table belongs INSIDE the activation
      float v = acc[e] + bv + (tab ? rv[e] : 0.f);
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (!tab) v += rv[e];
      fin[e] = v;
      acc[e] = 0.f;
    }
    ff_store_tile(cp, g.ldc, e_row0, e_col, g.M, e_colok, fin);
    if (MODE == 2)
      ff_emit_ln_stats(fin, lds + 3 * BUF_FLOATS + wave * (32 * 33), l32, half, e_row0 - 4 * half, g.M, g.ln_out,
                       g.N >> 5, (e_col - l32) >> 5);
  };

  // prologue: slices 0,1 -> LDS; slices 2,3 -> staging registers
  load_next(ra[0], rw[0], 0); advance(false);   // (nsl >= 4: slices 0..3 belong to the first tile)
  load_next(ra[1], rw[1], 1); advance(false);
  store_from(ra[0], rw[0], 0, 0);
  store_from(ra[1], rw[1], 1, 1);
  load_next(ra[0], rw[0], 0); advance(false);
  load_next(ra[1], rw[1], 1); advance(true);
  begin_tile(0);
  __syncthreads();
  read_frags(fa[0], fb[0], 0);

  int b0 = 0, b1 = 1, b2 = 2;
#ifdef FF_EXP_STAMP
  int _slice = 0;
#endif
  // Outer loop over this block's tiles, inner loop over the K-slices of one tile (two per trip: static register-set
  // indices).  The software pipeline (staging registers, LDS ring, load cursor) runs across the tile boundary; only
  // the epilogue sits between two inner loops, so the hot loop body is one straight basic block.
  for (cp_k = 0; cp_k < my_tiles; ++cp_k) {
    for (int s = 0; s < nsl; s += 2) {  // nsl is even (checked on the host)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (MODE == 1 && pend > 0 && --pend == 0) ln_finish();   // block-uniform
        read_frags(fa[u ^ 1], fb[u ^ 1], b1);
        store_from(ra[u], rw[u], b2, u);
        load_next(ra[u], rw[u], u);
        mfma_frags(fa[u], fb[u]);
        ff_persist_hints<MODE>();
        advance(true);
        FF_EXP_STAMP_PRE();
        __syncthreads();
        FF_EXP_STAMP_POST();
        { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
      }
    }
    end_tile();
    if (cp_k + 1 < my_tiles) begin_tile(cp_k + 1);
  }
}

template <int MODE>  // 0 plain, 1 LayerNorm-normalised A rows + row-indexed additive table, 2 emits row statistics of C
__global__ __launch_bounds__(256) void gemm_persist_kernel(GemmArgs g, int total_tiles) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  gemm_persist_body<MODE>(g, total_tiles, (int)blockIdx.x, (int)gridDim.x, lds);
}

// ---- stream-K form of the persistent kernel -----------------------------------------------------------------
// The persistent kernel hands out WHOLE tiles, so a launch whose tile count is not a multiple of the
// resident blocks pays a full extra round (288 tiles on 256 CUs cost 2 tile times, 32 tiles keep 224
// CUs idle while 32 blocks run the whole K chain).  Here the launch is cut into `unit`s of two K-slices
// and every block gets the same number of consecutive units of the flat (tile, unit) sequence, so a
// block's range is: the END part of its first tile, whole tiles, the BEGINNING part of its last tile.
//   * beginning part (tile not finished by this block): computed FIRST, raw accumulators written to
//     this block's 16 KB workspace slot, then flag[block] = 1;
//   * whole tiles: as in the persistent kernel;
//   * end part: computed LAST; the block owns the tile: it waits for the flags of the lower-numbered
//     blocks that hold the tile's earlier units (they published at the very start of their run, so the
//     wait is over before it begins), adds their partials in ascending block order (deterministic sum)
//     and runs the normal bias / activation / residual epilogue.
// A block therefore only ever waits on blocks that publish before doing anything else: no deadlock even
// when not all blocks are resident.  Every flagged slot has exactly one reader, which clears the flag
// after use: the workspace is clean again at the end of every launch (and a launch captured in a
// hipGraph replays correctly).
struct StreamK {
  float* ws;                   // [grid][16 accumulator registers][256 threads]
  unsigned int* flags;         // [grid]: 1 = slot holds a partial tile
  int upt;                     // units per tile = K / 64
  int base, rem;               // block lb owns base + (lb < rem) units
  // HYBRID launch (nA > 0; round 5): hw = tiles / CUs WHOLE tiles per CU, spread over TWO resident blocks per CU as in the
  // persistent kernel -- the first nA "heavy" blocks take tH = ceil(hw / 2) tiles each, the "light" blocks behind them tL =
  // floor(hw / 2) -- and the units of the remaining tiles (less than one tile per CU) are dealt in equal ranges to the first gx
  // light blocks, which run them BEFORE their whole tiles and exchange partial tiles among themselves.  A launch of 1.125 tiles per
  // CU (t = 9: 288 tiles of the 512-column projections) costs the CU 8 + 1 units instead of 9 + the exchange of EVERY tile (unit
  // ranges) or 16 (whole-tile rounds); one of 3.125 tiles per CU (t = 25) 3 tiles on two blocks + 2 units.
  int nA, tH, tL, uA, gx;      // uA = (nA * (tH + tL)) * upt: first unit of the dealt region
};

template <int MODE>  // as gemm_persist_kernel
__global__ __launch_bounds__(256) void gemm_streamk_kernel(GemmArgs g, StreamK sk) {
  constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 4, KF = BK / 8;
  constexpr int BUF_FLOATS = (BM + BN) * LDS_LD;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  const int wm0 = (wave >> 1) * 32, wn0 = (wave & 1) * 32;
  const int c4 = tid & 7, r = tid >> 3;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // logical block index: the blocks of one XCD (blockIdx % 8) own neighbouring unit ranges
  const int upt = sk.upt;
  int lb = 0, u0 = 0, u1 = 0;   // lb: index among the blocks that share unit ranges (slot of the partial-tile workspace)
  int main0 = 0, nmain = 0;     // hybrid launch: whole tiles [main0, main0 + nmain) of this block, run AFTER its unit range
  if (sk.nA > 0 && (int)blockIdx.x < sk.nA) {          // heavy block (block-uniform branches)
    const int GA = sk.nA;
    const int la = ((GA & 7) == 0) ? (blockIdx.x & 7) * (GA >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    main0 = la * sk.tH; nmain = sk.tH;
  } else {
    const int bb = blockIdx.x - sk.nA, G = gridDim.x - sk.nA;
    lb = ((G & 7) == 0) ? (bb & 7) * (G >> 3) + (bb >> 3) : bb;
    if (sk.nA == 0 || lb < sk.gx) {
      u0 = sk.uA + lb * sk.base + (lb < sk.rem ? lb : sk.rem);
      u1 = u0 + sk.base + (lb < sk.rem ? 1 : 0);
    }
    if (sk.nA > 0) { main0 = sk.nA * sk.tH + lb * sk.tL; nmain = sk.tL; }
  }
  const bool has_x = u0 < u1;
  if (!has_x && nmain == 0) return;
  FF_EXP_SKSTAMP(0);
  const int k0 = has_x ? u0 / upt : 0, k1 = has_x ? (u1 - 1) / upt : 0;
  const int ja = u0 - k0 * upt;  // first unit of tile k0 in the range
  const int jb = has_x ? u1 - k1 * upt : upt;  // one past the last unit of tile k1 in the range (1..upt)
  const bool has_c = has_x && jb < upt;                           // beginning part of k1: contributed
  const bool has_o = has_x && ja > 0 && !(k0 == k1 && has_c);     // end part of k0: owned, needs the fix-up
  const int kf0 = k0 + (ja > 0 ? 1 : 0);
  const int nfull = (has_x && (k1 + (has_c ? 0 : 1) - kf0) > 0) ? (k1 + (has_c ? 0 : 1) - kf0) : 0;
  const int nseg_x = (has_c ? 1 : 0) + nfull + (has_o ? 1 : 0);
  const int nseg = nseg_x + nmain;
  // segment p in execution order -> (tile, first slice, slice count, kind 0 whole / 1 contribute / 2 own+fix)
  // Order: the contributed part first (published at once), then every WHOLE tile (of the unit range, then the block's own tiles of
  // a hybrid launch), the owned part last.  A one-unit (two-slice) segment is therefore only ever the first or the last one: the
  // deferred LayerNorm statistics of MODE 1 need three slices between two tile switches of the load cursor.
  auto segment = [&](int p, int& tile, int& j0, int& n, int& kind) {
    if (has_c && p == 0) {
      tile = k1; j0 = 2 * (k1 == k0 ? ja : 0); n = 2 * jb - j0; kind = 1;
    } else {
      const int q = p - (has_c ? 1 : 0);
      if (q < nfull) { tile = kf0 + q; j0 = 0; n = nsl; kind = 0; }
      else if (q < nfull + nmain) { tile = main0 + (q - nfull); j0 = 0; n = nsl; kind = 0; }
      else { tile = k0; j0 = 2 * ja; n = nsl - j0; kind = 2; }
    }
  };

  // load cursor
  const float* a_ptr[2];
  const float* w_ptr[2];
  float cur_mu[2] = {0.f, 0.f}, cur_rs[2] = {1.f, 1.f};   // MODE 1: see gemm_persist_kernel
  LnRaw ln_raw[2];
  int pend = 0;
  auto ln_finish = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p) ff_ln_finish(ln_raw[p], g.ln_nseg, c4, g.ln_eps, cur_mu[p], cur_rs[p]);
  };
  auto set_load_tile = [&](int id, bool defer) {
    const int bz = id / tiles_mn, rem2 = id - bz * tiles_mn;
    const int m0 = (rem2 / g.tiles_n) * BM, n0 = (rem2 % g.tiles_n) * BN;
    const float* Asrc = ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + (long long)bz * g.batch_stride_a;
    const float* W = g.W + (long long)bz * g.batch_stride_w;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      int row = m0 + r + 32 * p;
      row = row < g.M ? row : g.M - 1;
      a_ptr[p] = Asrc + (size_t)row * g.lda + c4 * 4;
      if (MODE == 1) ff_ln_issue(g.ln_in + (size_t)row * g.ln_nseg * 2, g.ln_nseg, c4, ln_raw[p]);
      int n = n0 + r + 32 * p;
      n = n < g.N ? n : g.N - 1;
      w_ptr[p] = W + (size_t)n * g.ldw + c4 * 4;
    }
    if (MODE == 1) { if (defer) pend = 3; else ln_finish(); }
  };
  int ld_p = 0, ld_j, ld_end;
  {
    int tile, j0, n, kind;
    segment(0, tile, j0, n, kind);
    set_load_tile(tile, false);
    ld_j = j0; ld_end = j0 + n;
  }
  f32x4 ra[2][2], rw[2][2];
  auto load_next = [&](f32x4* xa, f32x4* xw, int u) {
    const int kk0 = ld_j * BK;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      xa[p] = *reinterpret_cast<const f32x4*>(a_ptr[p] + kk0);
      xw[p] = *reinterpret_cast<const f32x4*>(w_ptr[p] + kk0);
    }
  };
  auto advance = [&](bool defer) {  // block-uniform; past the last segment the cursor stays on its last slice
    if (++ld_j == ld_end) {
      if (ld_p + 1 < nseg) {
        int tile, j0, n, kind;
        segment(++ld_p, tile, j0, n, kind);
        set_load_tile(tile, defer);
        ld_j = j0; ld_end = j0 + n;
      } else {
        ld_j = ld_end - 1;
      }
    }
  };
  float* const st_a = lds + r * LDS_LD + c4 * 4;
  float* const st_w = st_a + BM * LDS_LD;
  auto store_from = [&](const f32x4* xa, const f32x4* xw, int buf, int u) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(st_a + buf * BUF_FLOATS + 32 * p * LDS_LD) =
          MODE == 1 ? (xa[p] - cur_mu[p]) * cur_rs[p] : xa[p];
      *reinterpret_cast<f32x4*>(st_w + buf * BUF_FLOATS + 32 * p * LDS_LD) = xw[p];
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

  // compute-side segment state
  int cp_p = 0, cp_n = 0, cp_kind = 0;
  int e_row0 = 0, e_col = 0;
  long long e_coff = 0;
  bool e_colok = false;
  float bv = 0.f, rv[16];
  auto begin_segment = [&](int p) {
    int id, j0;
    segment(p, id, j0, cp_n, cp_kind);
    if (cp_kind == 1) return;  // raw partial: no epilogue operands
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
    } else if (MODE == 1 && g.rowtab && colc < g.rowtab_cols) {  // additive table indexed by row / div (positions)
      ff_load_rowtab(g, e_row0, colc, rv);
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = 0.f;
    }
  };
  // Partials and flags cross XCDs (one L2 each).  An agent-scope release / acquire pair would write
  // back and invalidate the WHOLE L2 of both blocks (buffer_wbl2 / buffer_inv: measured 2-4x slower
  // launches), so only these few words are made coherent: relaxed agent-scope atomic stores / loads
  // (sc1 accesses that bypass the non-coherent cache levels), ordered by vmcnt(0) + the block barrier
  // on the writer and by the data dependence on the flag on the reader.
  auto end_segment = [&]() {
    FF_EXP_SKSTAMP(cp_kind == 1 ? 1 : (cp_kind == 2 ? 3 : 6));
    if (cp_kind == 1) {  // hand over the raw accumulators: slot[lb][4 quads][256 threads] float4
      // 16-byte write-through (sc1) stores: four fabric writes per thread instead of sixteen 4-byte ones (MI355X: a scalar
      // sc1 store is one fabric write whatever its width; inline asm: the compiler has no vector form of an agent-scope access)
      f32x4* wp = reinterpret_cast<f32x4*>(sk.ws + (size_t)lb * 4096) + tid;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + q * 256), "v"(v) : "memory");
        acc[4 * q] = 0.f; acc[4 * q + 1] = 0.f; acc[4 * q + 2] = 0.f; acc[4 * q + 3] = 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(sk.flags + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      FF_EXP_SKSTAMP(2);
      return;
    }
    if (cp_kind == 2) {  // add the partials of the blocks that hold units [k0 * upt, u0) of this tile
      const int ub = k0 * upt - sk.uA;   // (position inside the dealt region; whole-tile blocks never get here)
      const int big = sk.rem * (sk.base + 1);
      const int c0 = ub < big ? ub / (sk.base + 1) : sk.rem + (ub - big) / sk.base;
      for (int c = c0; c < lb; ++c) {
        while (__hip_atomic_load(sk.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const f32x4* rp = reinterpret_cast<const f32x4*>(sk.ws + (size_t)c * 4096) + tid;
        f32x4 t[4];   // the four loads in flight together: one memory round trip per contributor
#pragma unroll
        for (int q = 0; q < 4; ++q)
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[q]) : "v"(rp + q * 256) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] += t[e >> 2][e & 3];
      }
      __syncthreads();  // every thread is past its flag polls
      if (tid < lb - c0) __hip_atomic_store(sk.flags + c0 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      FF_EXP_SKSTAMP(4);
    }
    float* cp = g.C + e_coff;
    float fin[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {   // all values first, then the stores back to back: see gemm_persist_body
      const bool tab = MODE == 1 && g.rowtab != nullptr;   // a position table belongs INSIDE the activation
      float v = acc[e] + bv + (tab ? rv[e] : 0.f);
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (!tab) v += rv[e];
      fin[e] = v;
      acc[e] = 0.f;
    }
    ff_store_tile(cp, g.ldc, e_row0, e_col, g.M, e_colok, fin);
    if (MODE == 2)
      ff_emit_ln_stats(fin, lds + 3 * BUF_FLOATS + wave * (32 * 33), l32, half, e_row0 - 4 * half, g.M, g.ln_out,
                       g.N >> 5, (e_col - l32) >> 5);
  };

  // prologue: slices 0,1 -> LDS; slices 2,3 -> staging registers
  // Segments hold an even number (>= 2) of slices: a 2-slice first segment ends at the second advance -- the
  // statistics of the next tile are then merged on the spot (slices 0, 1 keep the first segment's, captured here).
  load_next(ra[0], rw[0], 0); advance(false);
  load_next(ra[1], rw[1], 1);
  const float p_mu0 = cur_mu[0], p_mu1 = cur_mu[1], p_rs0 = cur_rs[0], p_rs1 = cur_rs[1];
  advance(false);
  {
    const float n_mu0 = cur_mu[0], n_mu1 = cur_mu[1], n_rs0 = cur_rs[0], n_rs1 = cur_rs[1];
    cur_mu[0] = p_mu0; cur_mu[1] = p_mu1; cur_rs[0] = p_rs0; cur_rs[1] = p_rs1;
    store_from(ra[0], rw[0], 0, 0);
    store_from(ra[1], rw[1], 1, 1);
    cur_mu[0] = n_mu0; cur_mu[1] = n_mu1; cur_rs[0] = n_rs0; cur_rs[1] = n_rs1;
  }
  load_next(ra[0], rw[0], 0); advance(false);
  load_next(ra[1], rw[1], 1); advance(true);
  begin_segment(0);
  __syncthreads();
  read_frags(fa[0], fb[0], 0);

  int b0 = 0, b1 = 1, b2 = 2;
  // Outer loop over the segments of this block's unit range, inner loop over the slices of one segment (as in
  // gemm_persist_kernel: the pipeline state runs across, the hand-over / epilogue sits between two inner loops).
  for (cp_p = 0; cp_p < nseg; ++cp_p) {
    for (int s = 0; s < cp_n; s += 2) {  // segments hold an even number of slices
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (MODE == 1 && pend > 0 && --pend == 0) ln_finish();   // block-uniform
        read_frags(fa[u ^ 1], fb[u ^ 1], b1);
        store_from(ra[u], rw[u], b2, u);
        load_next(ra[u], rw[u], u);
        mfma_frags(fa[u], fb[u]);
        ff_persist_hints<MODE>();
        advance(true);
        __syncthreads();
        { const int tmp = b0; b0 = b1; b1 = b2; b2 = tmp; }
      }
    }
    end_segment();
    if (cp_p + 1 < nseg) begin_segment(cp_p + 1);
  }
  FF_EXP_SKSTAMP(5);
}

// hipFuncSetAttribute is per device: one flag per (kernel, device)
constexpr int FF_MAX_DEV = 16;
struct AttrFlags { std::atomic<bool> done[FF_MAX_DEV]; };   // idempotent attribute: host threads may race here
template <typename K>
int set_lds_limit(K kernel, int bytes, AttrFlags* fl) {
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  const bool track = dev >= 0 && dev < FF_MAX_DEV;
  if (!track || !fl->done[dev].load(std::memory_order_acquire)) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (track) fl->done[dev].store(true, std::memory_order_release);
  }
  return FF_OK;
}

template <int BM, int BN, int WM, int WN>
int launch_generic(GemmArgs g, int batch, hipStream_t st) {
  static AttrFlags attr_set = {};
  constexpr int bytes = 2 * (BM + BN) * 36 * (int)sizeof(float);
  FF_RETURN_IF(set_lds_limit(&gemm_generic_kernel<BM, BN, WM, WN>, bytes, &attr_set));
  g.tiles_m = ff_cdiv(g.M, BM);
  g.tiles_n = ff_cdiv(g.N, BN);
  hipLaunchKernelGGL((gemm_generic_kernel<BM, BN, WM, WN>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256), bytes,
                     st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

template <int BM, int BN, int WM, int WN, int BK = 32>
int launch_pipe(GemmArgs g, int batch, hipStream_t st) {
  if (g.K % BK != 0) return launch_generic<BM, BN, WM, WN>(g, batch, st);
  static AttrFlags attr_set = {};
  constexpr int bytes = 3 * (BM + BN) * (BK + 4) * (int)sizeof(float);
  FF_RETURN_IF(set_lds_limit(&gemm_pipe_kernel<BM, BN, WM, WN, BK>, bytes, &attr_set));
  g.tiles_m = ff_cdiv(g.M, BM);
  g.tiles_n = ff_cdiv(g.N, BN);
  hipLaunchKernelGGL((gemm_pipe_kernel<BM, BN, WM, WN, BK>), dim3(g.tiles_m * g.tiles_n, batch), dim3(256), bytes, st,
                     g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

inline int gemm_mode(const GemmArgs& g) { return g.ln_in ? 1 : (g.ln_out ? 2 : 0); }
constexpr int LN_PATCH_BYTES = 4 * 32 * 33 * (int)sizeof(float);  // MODE 2: one [32][33] patch per wave

template <int MODE>
int launch_persist_mode(GemmArgs g, int batch, hipStream_t st) {
  static AttrFlags attr_set = {};
  constexpr int bytes = 3 * 128 * 36 * (int)sizeof(float) + (MODE == 2 ? LN_PATCH_BYTES : 0);
  FF_RETURN_IF(set_lds_limit(&gemm_persist_kernel<MODE>, bytes, &attr_set));
  g.tiles_m = ff_cdiv(g.M, 64);
  g.tiles_n = ff_cdiv(g.N, 64);
  const long total = (long)g.tiles_m * g.tiles_n * batch;
  const int grid = total < 512 ? (int)total : 512;  // 256 CUs x 2 resident blocks
  hipLaunchKernelGGL(gemm_persist_kernel<MODE>, dim3(grid), dim3(256), bytes, st, g, (int)total);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

int launch_persist(GemmArgs g, int batch, hipStream_t st) {
  const int mode = gemm_mode(g);
  if (g.K % 64 != 0 || g.K < 128) {
    FF_CHECK_ARG(mode == 0, "ff_gemm_f32: the LayerNorm-fused forms need K %% 64 == 0 and K >= 128 (K=%d)", g.K);
    return launch_pipe<64, 64, 32, 32>(g, batch, st);
  }
  if (mode == 1) return launch_persist_mode<1>(g, batch, st);
  if (mode == 2) return launch_persist_mode<2>(g, batch, st);
  return launch_persist_mode<0>(g, batch, st);
}

// Stream-K workspace: one per (device, stream) -- launches on one stream are ordered, launches on
// different streams may overlap and must not share partial-tile slots.  Allocated on first use, kept
// for the life of the process (512 slots x 16 KB + flags).
constexpr int SK_MAX_GRID = 512;
struct SkWorkspace {
  int device;
  hipStream_t st;
  float* ws;
  unsigned int* flags;
};
std::mutex g_sk_mu;
std::vector<SkWorkspace> g_sk;
int g_sk_min_units = 2;      // smallest range handed to a block (units of 64 k)
int g_sk_two_per_cu = 2048;  // from this many units on, 512 blocks (2 per CU); below, at most 256
int g_small_max_rows = 1024;  // tile 7 hands launches with at most this many rows to gemm_small_kernel
double g_sk_fix_units = 2.5;  // what cutting tiles costs a launch, in units of per-CU work (policy only)

int sk_acquire(hipStream_t st, StreamK* out) {
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_sk_mu);
  for (SkWorkspace& w : g_sk)
    if (w.device == dev && w.st == st) {
      out->ws = w.ws; out->flags = w.flags;
      return FF_OK;
    }
  SkWorkspace w{dev, st, nullptr, nullptr};
  FF_CHECK_HIP(hipMalloc(&w.ws, (size_t)SK_MAX_GRID * 4096 * sizeof(float)));
  FF_CHECK_HIP(hipMalloc(&w.flags, SK_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipMemset(w.flags, 0, SK_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipDeviceSynchronize());
  g_sk.push_back(w);
  out->ws = w.ws; out->flags = w.flags;
  return FF_OK;
}

// ---- small-M kernel: one 32x32 output tile per block, K split over the four waves ---------------------------
// The first decode steps (and the whole seq2seq variant) launch products with a few hundred rows: every
// kernel above spends its time in fill (LDS staging, barrier, fragment read), drain, and -- when stream-K cuts
// the 64x64 tiles -- a cross-block exchange through memory (10-13 us per launch for < 2 us of MFMA work).
// Here nothing is staged: with K-contiguous operands a lane can load its MFMA operands straight from
// global memory (row = lane & 31, four consecutive k per 16-byte load; lane half h takes k = 8j + 4h .. +3 of
// every 8-wide group, the same for A and W), each wave accumulates a quarter of K for the same 32x32 tile,
// and the four partial tiles meet in 16 KB of LDS; every wave finishes four of the sixteen accumulator rows.
// NW waves share the K range of one 32x32 tile (KQ = K / NW each, a multiple of 32).  Four waves when the launch has
// enough tiles to fill the chip; EIGHT when it has at most one tile per CU anyway (the single-sequence decode, the
// last-layer / last-row launches): the dependent MFMA chain of a wave -- the longest serial piece of such a
// launch -- halves (K = 512: 64 -> 32 MFMAs, 1.7 -> 0.85 us).
template <int KQ, int MODE, int NW>  // MODE as gemm_persist_kernel
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float red[ff_gemm_small_lds_floats(MODE, NW)];
  ff_gemm_small_tile<KQ, MODE, NW>(g, (int)blockIdx.x, (long long)blockIdx.y, red);
}

// ---- panel form of the small-M kernel: ONE coalesced memory round trip per 512-wide K chunk -------------------------------
// gemm_small_kernel's lanes load their MFMA operands straight from global memory: 16 bytes of 32 different rows per instruction,
// i.e. 32 cache lines touched per instruction and every line touched by 4 instructions -- 4096 line requests per 32x32x512
// tile on the CU's one texture-address unit (~1.7 us of address processing), in two dependent batches (~1.5 us each: the
// second is requested after the first has been consumed).  A launch of at most one tile per CU is nothing but that chain.
// Here the eight waves of a block fetch the tile's A panel [32 x KC] and W panel [32 x KC] (KC <= 512 columns) with fully
// coalesced 16-byte loads -- every line requested once, ALL requests of the chunk in flight together -- write them to LDS
// (rows padded by 4 floats: conflict-free ds_read_b128 of the MFMA fragments), and wave w then multiplies its eighth of
// the chunk exactly like gemm_small_kernel does (same k order inside a wave, same partial-tile reduction: same result bits).
// K = 1024 runs two chunks.  LDS: 2 x 32 x 516 floats = 129 KB for KC = 512 -> one block per CU: used where the small-M
// kernel ran its eight-wave form (at most one tile per CU).
template <int KC, int MODE>  // MODE as gemm_persist_kernel
__global__ __launch_bounds__(512) void gemm_panel_kernel(GemmArgs g) {
  constexpr int NW = 8, LD = KC + 4, KQ = KC / NW, NG = KQ / 8;
  constexpr int NV = KC / 64;             // float4 per thread and panel
  constexpr int RPW = 16 / NW;            // accumulator registers a wave finishes
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const As = lds;                  // [32][LD]
  float* const Ws = lds + 32 * LD;        // [32][LD]
  float* const lnrow = lds + 64 * LD;     // MODE 1: [32][2] (mean, rstd)
  float* const red = lds;                 // after the MFMA chain: [8][16][64] partial tiles (+ MODE 2 patch [32][33])
  static_assert(64 * LD >= NW * 16 * 64 + 32 * 33, "the partial tiles reuse the panel area");
  FF_EXP_PSTAMP(0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int m0 = (blockIdx.x / g.tiles_n) * 32, n0 = (blockIdx.x % g.tiles_n) * 32;
  const long long bz = blockIdx.y;
  const float* Asrc = ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  const float* Wsrc = g.W + bz * g.batch_stride_w;
  // staging map: float4 i of this thread is chunk column c4 = (tid + 512 i) % (KC/4) of panel row (tid + 512 i) / (KC/4):
  // a wave's instruction covers 1 KB of consecutive addresses of one row
  constexpr int C4 = KC / 4;
  const float* a_ptr[NV];
  const float* w_ptr[NV];
  int srow[NV];
  int soff[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int idx = tid + 512 * i;
    const int r = idx / C4, c4 = idx % C4;
    int row = m0 + r, col = n0 + r;
    row = row < g.M ? row : g.M - 1;
    col = col < g.N ? col : g.N - 1;
    a_ptr[i] = Asrc + (size_t)row * g.lda + c4 * 4;
    w_ptr[i] = Wsrc + (size_t)col * g.ldw + c4 * 4;
    srow[i] = r;
    soff[i] = r * LD + c4 * 4;
  }
  f32x4 ra[NV], rw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i]);
    rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i]);
  }
  // epilogue operands and (MODE 1) row statistics ride in the same round trip
  f32x4 sv = {0.f, 0.f, 0.f, 0.f};
  const int spart = lane & 7, strow = wave * (32 / NW) + ((lane >> 3) % (32 / NW));
  if (MODE == 1) {
    int r = m0 + strow;
    r = r < g.M ? r : g.M - 1;
    if (2 * spart < g.ln_nseg) sv = *reinterpret_cast<const f32x4*>(g.ln_in + ((size_t)r * g.ln_nseg + 2 * spart) * 2);
  }
  const int ocol = n0 + l32;
  const bool colok = ocol < g.N;
  const float bv = (g.bias && colok) ? g.bias[ocol] : 0.f;
  const bool tab = MODE == 1 && g.rowtab != nullptr && !g.res;
  float rv[RPW];
  int orow[RPW], prow[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int e = wave * RPW + q;
    prow[q] = (e & 3) + 8 * (e >> 2) + 4 * half;
    orow[q] = m0 + prow[q];
    const bool ok = colok && orow[q] < g.M;
    rv[q] = 0.f;
    if (g.res) { if (ok) rv[q] = g.res[bz * g.batch_stride_c + (size_t)orow[q] * g.ldr + ocol]; }
    else if (tab && ok && ocol < g.rowtab_cols)
      rv[q] = g.rowtab[(size_t)(orow[q] / g.rowtab_div) * g.ld_rowtab + ocol];
  }
  if (MODE == 1) {
    // Chan's update over the row's 32-column segments, two per lane, eight lanes per row (ln_nseg even, <= 16)
    const bool sok = 2 * spart < g.ln_nseg;
    const float fn = (float)g.ln_nseg;
    const float mean = ff_sum8(sok ? sv.x + sv.z : 0.f) / fn;
    const float m2 = ff_sum8(sok ? sv.y + sv.w : 0.f);
    const float d0 = sv.x - mean, d1 = sv.z - mean;
    const float dev = ff_sum8(sok ? d0 * d0 + d1 * d1 : 0.f);
    const float var = (m2 + 32.f * dev) / (32.f * fn);
    if (spart == 0 && lane < 8 * (32 / NW)) *reinterpret_cast<f32x2*>(lnrow + 2 * strow) = f32x2{mean, 1.0f / sqrtf(var + g.ln_eps)};
    __syncthreads();
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int nchunks = g.K / KC;
  for (int ch = 0; ch < nchunks; ++ch) {
    // ---- registers -> LDS (A normalised on the way in MODE 1) ----
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x4 av = ra[i];
      if (MODE == 1) {
        const f32x2 ms = *reinterpret_cast<const f32x2*>(lnrow + 2 * srow[i]);
        av = (av - ms.x) * ms.y;
      }
      *reinterpret_cast<f32x4*>(As + soff[i]) = av;
      *reinterpret_cast<f32x4*>(Ws + soff[i]) = rw[i];
    }
    if (ch + 1 < nchunks) {   // the next chunk's requests leave before this chunk is consumed
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(a_ptr[i] + (ch + 1) * KC);
        rw[i] = *reinterpret_cast<const f32x4*>(w_ptr[i] + (ch + 1) * KC);
      }
    }
    __syncthreads();
    FF_EXP_PSTAMP(1);
    // ---- wave w: columns [w KQ, w KQ + KQ) of the chunk; lane half h takes k = 8j + 4h .. +3 of every 8-wide group ----
    const float* fa = As + l32 * LD + wave * KQ + half * 4;
    const float* fb = Ws + l32 * LD + wave * KQ + half * 4;
    f32x4 a[NG], b[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      a[j] = *reinterpret_cast<const f32x4*>(fa + j * 8);
      b[j] = *reinterpret_cast<const f32x4*>(fb + j * 8);
    }
#pragma unroll
    for (int j = 0; j < NG; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][c], b[j][c], acc, 0, 0, 0);
    __syncthreads();   // the panels are overwritten by the next chunk / the partial tiles
    FF_EXP_PSTAMP(2);
  }
  // partial tiles -> LDS [wave][reg][lane]; wave w then finishes registers RPW*w .. RPW*w + RPW-1
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 16 + e) * 64 + lane] = acc[e];
  __syncthreads();
  float* Cout = g.C + bz * g.batch_stride_c;
  float v[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int e = wave * RPW + q;
    v[q] = (red[(0 * 16 + e) * 64 + lane] + red[(1 * 16 + e) * 64 + lane]) +
           (red[(2 * 16 + e) * 64 + lane] + red[(3 * 16 + e) * 64 + lane]);
    v[q] += (red[(4 * 16 + e) * 64 + lane] + red[(5 * 16 + e) * 64 + lane]) +
            (red[(6 * 16 + e) * 64 + lane] + red[(7 * 16 + e) * 64 + lane]);
  }
  float* patch = red + NW * 16 * 64;
  float o[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {   // values first, stores behind them (see gemm_persist_body)
    o[q] = v[q] + bv + (tab ? rv[q] : 0.f);
    if (g.act == 1) o[q] = fmaxf(o[q], 0.f);
    if (!tab) o[q] += rv[q];
    if (MODE == 2) patch[prow[q] * 33 + l32] = o[q];
  }
  FF_EXP_PSTAMP(3);
#pragma unroll
  for (int q = 0; q < RPW; ++q)
    if (colok && orow[q] < g.M) Cout[(size_t)orow[q] * g.ldc + ocol] = o[q];
  FF_EXP_PSTAMP(4);
  if (MODE == 2) {  // row statistics of the finished 32x32 tile: 64 threads, (row, column half) each
    __syncthreads();
    if (tid < 64) {
      float x[16], sm = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { x[c] = patch[l32 * 33 + half * 16 + c]; sm += x[c]; }
      sm = ff_halves_sum(sm);
      const float mean = sm * (1.0f / 32.0f);
      float m2 = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) { const float d = x[c] - mean; m2 += d * d; }
      m2 = ff_halves_sum(m2);
      const int r = m0 + l32;
      if (half == 0 && r < g.M) *reinterpret_cast<f32x2*>(g.ln_out + ((size_t)r * (g.N >> 5) + (n0 >> 5)) * 2) = f32x2{mean, m2};
    }
  }
}

template <int KC, int MODE>
int launch_panel(const GemmArgs& g, dim3 grid, hipStream_t st) {
  static AttrFlags attr_set = {};
  constexpr int bytes = (64 * (KC + 4) + 64) * (int)sizeof(float);
  FF_RETURN_IF(set_lds_limit(&gemm_panel_kernel<KC, MODE>, bytes, &attr_set));
  hipLaunchKernelGGL((gemm_panel_kernel<KC, MODE>), grid, dim3(512), bytes, st, g);
  return FF_OK;
}

#ifndef FF_SMALL_WIDE_BLOCKS
#define FF_SMALL_WIDE_BLOCKS 256
#endif
int g_small_wide_blocks = FF_SMALL_WIDE_BLOCKS;  // launches with at most this many tiles split K over eight waves instead of four

inline int small_panel() { return ff_knob(FF_K_NO_PANEL) ? 0 : 1; }   // 1: launches of at most one tile per CU take gemm_panel_kernel (0: the eight-wave gemm_small_kernel)

template <int MODE>
int launch_small_mode(const GemmArgs& g, dim3 grid, hipStream_t st) {
  const bool wide = (long)grid.x * grid.y <= g_small_wide_blocks;
  const bool panel = wide && small_panel() != 0;
  switch (g.K) {
    case 512:
      if (panel) FF_RETURN_IF((launch_panel<512, MODE>(g, grid, st)));
      else if (wide) hipLaunchKernelGGL((gemm_small_kernel<64, MODE, 8>), grid, dim3(512), 0, st, g);
      else hipLaunchKernelGGL((gemm_small_kernel<128, MODE, 4>), grid, dim3(256), 0, st, g);
      break;
    case 1024:
      if (panel) FF_RETURN_IF((launch_panel<512, MODE>(g, grid, st)));
      else if (wide) hipLaunchKernelGGL((gemm_small_kernel<128, MODE, 8>), grid, dim3(512), 0, st, g);
      else hipLaunchKernelGGL((gemm_small_kernel<256, MODE, 4>), grid, dim3(256), 0, st, g);
      break;
    case 128: hipLaunchKernelGGL((gemm_small_kernel<32, MODE, 4>), grid, dim3(256), 0, st, g); break;
    case 256:
      if (panel) FF_RETURN_IF((launch_panel<256, MODE>(g, grid, st)));
      else if (wide) hipLaunchKernelGGL((gemm_small_kernel<32, MODE, 8>), grid, dim3(512), 0, st, g);
      else hipLaunchKernelGGL((gemm_small_kernel<64, MODE, 4>), grid, dim3(256), 0, st, g);
      break;
    default: ff_set_error("ff_gemm_f32: small-M kernel supports K in {128, 256, 512, 1024}"); return FF_ERR_ARG;
  }
  FF_CHECK_LAUNCH();
  return FF_OK;
}

int launch_small(GemmArgs g, int batch, hipStream_t st) {
  g.tiles_m = ff_cdiv(g.M, 32);
  g.tiles_n = ff_cdiv(g.N, 32);
  const dim3 grid(g.tiles_m * g.tiles_n, batch);
  const int mode = gemm_mode(g);
  if (mode == 1) return launch_small_mode<1>(g, grid, st);
  if (mode == 2) return launch_small_mode<2>(g, grid, st);
  return launch_small_mode<0>(g, grid, st);
}
bool small_ok(const GemmArgs& g) {
  return (g.K == 128 || g.K == 256 || g.K == 512 || g.K == 1024) && (!g.A2 || (g.n_split % 32) == 0);
}

// mode 0: whole tiles (persistent kernel) or equal unit ranges, whichever the cost model prefers; 2: unit ranges
template <int MODE>
int launch_streamk_mode(const GemmArgs& g, const StreamK& sk, int grid, hipStream_t st) {
  static AttrFlags attr_set = {};
  constexpr int bytes = 3 * 128 * 36 * (int)sizeof(float) + (MODE == 2 ? LN_PATCH_BYTES : 0);
  FF_RETURN_IF(set_lds_limit(&gemm_streamk_kernel<MODE>, bytes, &attr_set));
  hipLaunchKernelGGL(gemm_streamk_kernel<MODE>, dim3(grid), dim3(256), bytes, st, g, sk);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

int launch_streamk(GemmArgs g, int batch, hipStream_t st, int mode) {
  if (g.K % 64 != 0 || g.K < 128) return launch_persist(g, batch, st);   // (reports the LN-fused restriction)
  g.tiles_m = ff_cdiv(g.M, 64);
  g.tiles_n = ff_cdiv(g.N, 64);
  StreamK sk;
  sk.upt = g.K / 64;
  const long tiles = (long)g.tiles_m * g.tiles_n * batch;
  const long units = tiles * sk.upt;
  FF_CHECK_ARG(units < (1L << 30), "ff_gemm_f32: problem too large for the stream-K launcher");
  // Whole tiles (no exchange) when they spread evenly enough over the 256 CUs, otherwise equal unit
  // ranges: per-CU cost in units, the cut costing about g_sk_fix_units on top of the even share.
  const long cus = SK_MAX_GRID / 2;
  const double whole_cost = (double)((tiles + cus - 1) / cus) * sk.upt;
  const double split_cost = (double)units / cus + g_sk_fix_units;
  sk.nA = 0; sk.tH = 0; sk.tL = 0; sk.uA = 0; sk.gx = 0;
  // Hybrid (see StreamK): hw whole tiles per CU + the units of the remaining tiles dealt to a second block per CU.
  const int hyb_on = ff_knob(FF_K_SK_HYBRID);               // (A/B knob)
  const double hyb_fix = 0.1 * ff_knob(FF_K_SK_HYBRID_FIX);
  const long hw = tiles / cus, left = tiles % cus;
  // ... when at most half a round of tiles is left over: measured (profiles/r05/gemm_hybrid_ab.txt, 256 t rows) +15-26 % on the
  // 512-column projections at t = 9 / 10, +8-12 % at t = 12, -2...-4 % at t = 14 (0.75 rounds left: unit ranges stay).
  const long hyb_max_left8 = ff_knob(FF_K_SK_HYBRID_MAXLEFT8);   // (A/B knob: eighths of a round)
  const int hyb_force = ff_knob(FF_K_SK_HYBRID_FORCE);   // (probe: whole rounds too)
  // (the hybrid plan -- 256 heavy + up to 256 light blocks, light blocks waiting for partials of lower-numbered ones -- is made for
  //  and measured on the 256 CUs of an MI355X in SPX mode; a partition with another CU count keeps the older launch shapes)
  if (mode == 0 && hyb_on && ff_num_cus() == (int)cus && hw >= 1 && (left > 0 || hyb_force) && 8 * left <= hyb_max_left8 * cus) {
    const long left_units = left * sk.upt;
    const long hyb_min_units = ff_knob(FF_K_SK_HYBRID_MINU);   // (A/B knob)
    long gb = left_units / (hyb_min_units > 0 ? hyb_min_units : 1);
    if (gb > cus) gb = cus;
    if (gb < 1) gb = 1;   // (left == 0 under the probe knob: one light block with an empty range)
    const double hybrid_cost = (double)hw * sk.upt + (double)((left_units + gb - 1) / gb) + hyb_fix;
    if ((hybrid_cost < whole_cost && hybrid_cost < split_cost) || hyb_force) {
      sk.nA = (int)cus; sk.tH = (int)((hw + 1) / 2); sk.tL = (int)(hw / 2); sk.uA = (int)(cus * hw * sk.upt);
      sk.gx = (int)gb;
      sk.base = (int)(left_units / gb);
      sk.rem = (int)(left_units % gb);
      FF_RETURN_IF(sk_acquire(st, &sk));
      const int lmh = gemm_mode(g);
      const int gridh = (int)(cus + (sk.tL > 0 ? cus : gb));   // light blocks: all 256 when they carry whole tiles
      if (lmh == 1) return launch_streamk_mode<1>(g, sk, gridh, st);
      if (lmh == 2) return launch_streamk_mode<2>(g, sk, gridh, st);
      return launch_streamk_mode<0>(g, sk, gridh, st);
    }
  }
  if (mode == 0 && whole_cost <= split_cost) return launch_persist(g, batch, st);
  long grid;
  if (units >= g_sk_two_per_cu) grid = SK_MAX_GRID;
  else {
    grid = ff_cdiv((int)units, g_sk_min_units);
    if (grid > cus) grid = cus;
  }
  if (grid > units) grid = units;
  sk.base = (int)(units / grid);
  sk.rem = (int)(units % grid);
  FF_RETURN_IF(sk_acquire(st, &sk));
  const int lm = gemm_mode(g);
  if (lm == 1) return launch_streamk_mode<1>(g, sk, (int)grid, st);
  if (lm == 2) return launch_streamk_mode<2>(g, sk, (int)grid, st);
  return launch_streamk_mode<0>(g, sk, (int)grid, st);
}

}  // namespace

extern "C" int ff_gemm_prepare_stream(ff_stream_t stream) {
  StreamK sk;
  FF_RETURN_IF(sk_acquire((hipStream_t)stream, &sk));
  return ff_x3_prepare_stream((hipStream_t)stream);
}

extern "C" int ff_set_gemm_tuning(int min_units, int two_per_cu_units, int fix_tenths, int small_max_rows) {
  FF_CHECK_ARG(min_units >= 1 && two_per_cu_units >= 1 && fix_tenths >= 0 && small_max_rows >= 0,
               "ff_set_gemm_tuning: bad arguments");
  std::lock_guard<std::mutex> lock(g_sk_mu);
  g_sk_min_units = min_units;
  g_sk_two_per_cu = two_per_cu_units;
  g_sk_fix_units = 0.1 * fix_tenths;
  g_small_max_rows = small_max_rows;
  ff_tuning_changed();
  return FF_OK;
}

namespace {
// rows from which the automatic choice hands a launch to the LDS-DMA kernel (FF_DMA_MIN_ROWS overrides: probes).  Measured
// (profiles/r04/dma_threshold_ab.txt, gemm_dma_f32.txt): +3-8 % over the 64x64 / 128x64 families from ~8 k rows, LayerNorm-folded
// forms +6-12 % from ~9 k rows, equal around 4-6 k, slower below; config B 59.9 -> 59.4 ms, 128 wireframes per call 206 -> 214 k/s
// (with the LayerNorms folded at every size, which the 128x64 kernel could not).
inline long dma_min_rows() { return ff_knob(FF_K_DMA_MIN_ROWS); }
// ... and for the 512-column projections (out-proj, cross-q, linear2): 64 x 128 tiles give them only 4 tile columns, and since the
// 64x64 family has the hybrid launch (round 5) it is 3-14 % faster on them up to ~7.5 k rows (profiles/r05/gemm_families_4k_9k.txt:
// LayerNorm-folded forms, 4096 ... 6400 rows), equal at 7680, slower from 8448.
inline long dma_min_rows_n512() { return ff_knob(FF_K_DMA_MIN_ROWS_N512); }
// ... and for the 1536-column q|k|v projection (12 tile columns): its LayerNorm-folded form wins from ~2.5 k rows on (same file:
// 89.7 / 91.4 / 91.0 / 94.4 against 79.5 / 79.7 / 81.6 / 85.4 TF/s at 2816 / 3072 / 3584 / 3840 rows, equal at 2304 / 2560 / 3328).
inline long dma_min_rows_wide() { return ff_knob(FF_K_DMA_MIN_ROWS_WIDE); }
int gemm_dispatch(GemmArgs g, int batch, int tile, hipStream_t st) {
  const bool split128 = !g.A2 || (g.n_split % 128) == 0;
  if (tile == 0) tile = 7;  // stream-K kernel, launch shape by cost model (falls back by itself for K tails)
  if (tile == 5 && !split128) tile = 4;
  const int M = g.M, N = g.N, K = g.K;
  FFProfScope prof(FF_CAT_GEMM, 2.0 * M * N * K * batch, st);
  ff_prof_add_bytes(FF_CAT_GEMM, 4.0 * batch * ((double)M * K + (double)N * K + (double)M * N * (g.res ? 2 : 1)));
  // tile 11 / automatic from g_dma_min_rows rows on: the LDS-DMA kernel (ff_gemm_x3.hip: gemm_dma_f32_kernel)
  const bool dma_ok = ff_gemm_dma_f32_ok(g, batch);
  if (tile == 11) {
    FF_CHECK_ARG(dma_ok, "ff_gemm_f32: tile 11 needs K %% 32 == 0, K >= 64, N %% 4 == 0, leading dimensions %% 4, 16-byte aligned operands, batch 1");
    return ff_gemm_dma_f32(g, st);
  }
  if (tile == 12) {   // the same kernel with 64 x 64 tiles (round 6)
    FF_CHECK_ARG(dma_ok && (!g.A2 || (g.n_split % 64) == 0), "ff_gemm_f32: tile 12 needs what tile 11 needs");
    return ff_gemm_dma_f32(g, st, 64);
  }
  const long dmr = dma_min_rows(), dmr512 = dma_min_rows_n512(), dmrw = dma_min_rows_wide();
  const long dma_from = N <= 512 ? (dmr512 > dmr ? dmr512 : dmr) : (N >= 1536 && dmrw < dmr ? dmrw : dmr);
  if (tile == 7 && dma_ok && (long)M >= dma_from) return ff_gemm_dma_f32(g, st);
  switch (tile) {
    case 1: return launch_generic<64, 64, 32, 32>(g, batch, st);
    case 2: return launch_pipe<64, 64, 32, 32>(g, batch, st);
    case 3: return launch_persist(g, batch, st);
    case 4: return launch_pipe<128, 64, 64, 32>(g, batch, st);
    case 6: return launch_streamk(g, batch, st, 2);
    case 7:
      // few rows, the path's K (512 / 1024): the unstaged split-K kernel (its unshared operand loads cost more
      // than the staging from ~1000 rows on; with K < 512 a wave's share of K is too short to be worth it)
      // (wide outputs leave it earlier: at 1024 rows the persistent kernel is 8 % / 22 % faster for N = 1536 / 1024)
      if ((long)M * batch <= (N <= 512 ? g_small_max_rows : (g_small_max_rows * 3) / 4) && K >= 512 && small_ok(g))
        return launch_small(g, batch, st);
      // Plain projections of the large steps: 128x64 block tiles with 64x32 wave tiles and 16-wide slices (two
      // accumulators share every W fragment: 6 instead of 8 fragment reads per 16 MFMAs, half the global -> LDS
      // traffic per flop, still two blocks per CU) run 5-10 % faster than the 64x64 stream-K kernel -- as plain
      // one-tile blocks (a stream-K form of the same geometry was built and measured: no faster than the 64x64 one),
      // so only when the tile count fills the 512 resident block slots evenly (measured crossover: ~0.9).
      if (batch == 1 && M >= 4096 && (K & 15) == 0 && gemm_mode(g) == 0 && (!g.A2 || (g.n_split & 63) == 0)) {
        const long t9 = (long)ff_cdiv(M, 128) * ff_cdiv(N, 64);
        const long rounds = (t9 + 511) / 512;
        if (10 * t9 >= 9 * rounds * 512) return launch_pipe<128, 64, 64, 32, 16>(g, batch, st);
      }
      return launch_streamk(g, batch, st, 0);
    case 8: FF_CHECK_ARG(small_ok(g), "ff_gemm_f32: tile 8 needs K in {128,256,512,1024}"); return launch_small(g, batch, st);
    case 9: return launch_pipe<128, 64, 64, 32, 16>(g, batch, st);
    case 10: return launch_pipe<128, 128, 64, 64, 16>(g, batch, st);
    default: return launch_pipe<128, 128, 64, 64>(g, batch, st);
  }
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
#ifndef FF_EXP_NO_LD_CHECK   // (tools/gemm_latency_probe.py aliases all rows onto one: timing experiments only)
  FF_CHECK_ARG((lda & 3) == 0 && (ldw & 3) == 0 && lda >= K && ldw >= K && ldc >= N,
               "ff_gemm_f32: bad leading dimensions lda=%d ldw=%d ldc=%d", lda, ldw, ldc);
#endif
  FF_CHECK_ARG(ff_aligned16(A) && ff_aligned16(W) && (!A2 || ff_aligned16(A2)),
               "ff_gemm_f32: A/A2/W must be 16-byte aligned");
  FF_CHECK_ARG(!residual || ldr >= N, "ff_gemm_f32: bad ldr");
  FF_CHECK_ARG(act == 0 || act == 1, "ff_gemm_f32: act must be 0 or 1");
  FF_CHECK_ARG(tile >= 0 && tile <= 12, "ff_gemm_f32: tile must be 0..12");
  FF_CHECK_ARG(batch > 0 && batch <= 65535 && (stride_a & 3) == 0 && (stride_w & 3) == 0,
               "ff_gemm_f32: bad batch arguments");
  FF_CHECK_ARG(batch == 1 || !residual, "ff_gemm_f32: residual is not supported with batch > 1");
  if (A2) FF_CHECK_ARG(n_split > 0 && n_split < N && (n_split % 64) == 0, "ff_gemm_f32: n_split must be a multiple of 64 inside (0,N)");
  GemmArgs g{A, A2, W, bias, residual, C, lda, ldw, ldr, ldc, M, N, K, A2 ? n_split : N, act, 0, 0,
             stride_a, stride_w, stride_c, nullptr, 0, 0.f, nullptr, 0, 1, 0, nullptr};
  return gemm_dispatch(g, batch, tile, (hipStream_t)stream);
}

extern "C" int ff_gemm_f32_ln(const ff_gemm_ln_desc* d, ff_stream_t stream) {
  FF_CHECK_ARG(d != nullptr, "ff_gemm_f32_ln: null descriptor");
  const int M = d->M, N = d->N, K = d->K;
  if (M == 0 || N == 0) return FF_OK;
  FF_CHECK_ARG(M > 0 && N > 0 && K > 0 && (K & 3) == 0, "ff_gemm_f32_ln: bad M=%d N=%d K=%d (K %% 4)", M, N, K);
  FF_CHECK_ARG(d->A && d->W && d->C, "ff_gemm_f32_ln: null operand");
  FF_CHECK_ARG((d->lda & 3) == 0 && (d->ldw & 3) == 0 && d->lda >= K && d->ldw >= K && d->ldc >= N,
               "ff_gemm_f32_ln: bad leading dimensions lda=%d ldw=%d ldc=%d", d->lda, d->ldw, d->ldc);
  FF_CHECK_ARG(ff_aligned16(d->A) && ff_aligned16(d->W), "ff_gemm_f32_ln: A/W must be 16-byte aligned");
  FF_CHECK_ARG(!d->residual || d->ldr >= N, "ff_gemm_f32_ln: bad ldr");
  FF_CHECK_ARG(d->act == 0 || d->act == 1, "ff_gemm_f32_ln: act must be 0 or 1");
  FF_CHECK_ARG(d->tile == 0 || d->tile == 3 || d->tile == 6 || d->tile == 7 || d->tile == 8 || d->tile == 11 || d->tile == 12,
               "ff_gemm_f32_ln: tile must be 0, 3, 6, 7, 8 or 11 (the kernels that carry the LayerNorm fusion)");
  FF_CHECK_ARG(!(d->ln_stats_in && d->ln_stats_out), "ff_gemm_f32_ln: statistics in AND out in one launch are not supported");
  FF_CHECK_ARG(!d->ln_stats_in || (d->ln_nseg > 0 && d->ln_nseg * 32 == K && d->ln_eps >= 0.f),
               "ff_gemm_f32_ln: ln_nseg * 32 must equal K (the statistics describe whole rows of A)");
  FF_CHECK_ARG(!d->ln_stats_in || d->ln_nseg <= 8 * LN_NQ, "ff_gemm_f32_ln: ln_stats_in supports K <= %d", 256 * LN_NQ);
  FF_CHECK_ARG(!d->row_table || (d->ln_stats_in && !d->residual && d->row_div > 0 && d->row_cols > 0 &&
                                 d->row_cols <= N && d->ld_row_table >= d->row_cols),
               "ff_gemm_f32_ln: row_table needs ln_stats_in, no residual, row_div > 0 and 0 < row_cols <= N");
  FF_CHECK_ARG(!d->ln_stats_out || (N & 31) == 0, "ff_gemm_f32_ln: ln_stats_out needs N %% 32 == 0");
  FF_CHECK_ARG(!(d->ln_stats_in || d->ln_stats_out) || (K % 64 == 0 && K >= 128),
               "ff_gemm_f32_ln: the fused forms need K %% 64 == 0 and K >= 128");
  GemmArgs g{d->A, nullptr, d->W, d->bias, d->residual, d->C, d->lda, d->ldw, d->ldr, d->ldc, M, N, K, N, d->act, 0, 0,
             0, 0, 0, d->ln_stats_in, d->ln_nseg, d->ln_eps, d->row_table, d->ld_row_table, d->row_div > 0 ? d->row_div : 1,
             d->row_cols, d->ln_stats_out};
  return gemm_dispatch(g, 1, d->tile, (hipStream_t)stream);
}

extern "C" int ff_gemm_f32(const float* A, int lda, const float* A2, int n_split, const float* W,
                           int ldw, const float* bias, const float* residual, int ldr, float* C,
                           int ldc, int M, int N, int K, int act, int tile, ff_stream_t stream) {
  return ff_gemm_f32_batched(A, lda, A2, n_split, W, ldw, bias, residual, ldr, C, ldc, M, N, K, act, tile,
                             1, 0, 0, 0, stream);
}

