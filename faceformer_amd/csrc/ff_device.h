// Device-side building blocks shared by the kernels of the library (one launch per operator): small typed access helpers,
// the projection argument block, the small-M projection tile, the wave-per-unit attention block, the LayerNorm row and the
// pointer head's row reduction.  (Until round 4 every helper also had an agent-coherent form for the persistent "chain"
// launches of round 3; those were measured slower three ways -- DESIGN.md 8 -- and are gone.)
#pragma once
#include "ff_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FF_GLOBAL __attribute__((address_space(1)))
typedef unsigned long long ff_u64;

__device__ __forceinline__ f32x4 ff_ld16(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x2 ff_ld8(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ float ff_ld4(const float* p) { return *p; }
__device__ __forceinline__ void ff_st16(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void ff_st8(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ void ff_st4(float* p, float v) { *p = v; }
// (Round 6: agent-scope `sc1` write-through forms of these result stores -- the lines leave the XCD's L2 while the kernel runs
//  instead of in the end-of-kernel release -- were built and measured: config B 59.1-59.2 ms against 58.0-58.4; not kept,
//  profiles/r06/wt_stores_ab.txt.)
__device__ __forceinline__ int ff_ld4i(const int* p) { return *p; }
__device__ __forceinline__ void ff_st4i(int* p, int v) { *p = v; }
// The 16 accumulator rows of a lane (row0 + (e&3) + 8*(e>>2), one column) -> C.  Whole tiles -- the wave's 32 rows inside M --
// take 16 unguarded stores in a row; only the last row tile of a launch takes the guarded form.
__device__ __forceinline__ void ff_store_tile(float* cp, int ldc, int row0, int col, int M, bool colok, const float (&v)[16]) {
  float* p = cp + (size_t)row0 * ldc + col;
  if (row0 - (row0 & 4) + 32 <= M) {   // wave-uniform (row0 differs by 4 between the lane halves): rows row0 .. row0 + 27 exist
    if (colok) {
#pragma unroll
      for (int e = 0; e < 16; ++e) ff_st4(p + (size_t)((e & 3) + 8 * (e >> 2)) * ldc, v[e]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      if (row0 + (e & 3) + 8 * (e >> 2) < M && colok) ff_st4(p + (size_t)((e & 3) + 8 * (e >> 2)) * ldc, v[e]);
  }
}

__device__ __forceinline__ f32x4 ff_ldw16(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <typename T>
__device__ __forceinline__ T ff_ldw(const T* p) { return *p; }

// ---- projection kernels: argument block ------------------------------------------------------------------------------
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
  long long batch_stride_a, batch_stride_w, batch_stride_c;  // per-problem offsets (elements)
  // LayerNorm fusion (persistent / stream-K / small-M kernels only; batch == 1):
  const float* ln_in;   // MODE 1: [M][ln_nseg][2] (mean, M2 over 32 columns) segment statistics of the A rows
  int ln_nseg;
  float ln_eps;
  const float* rowtab;  // MODE 1: C[m][n] += rowtab[(m / rowtab_div) * ld_rowtab + n] for n < rowtab_cols (no residual then)
  int ld_rowtab, rowtab_div, rowtab_cols;
  float* ln_out;        // MODE 2: [M][N/32][2] segment statistics of the stored C rows
};

// the LDS-DMA f32 projection kernel (ff_gemm_x3.hip): eligibility of a launch and the launch itself
bool ff_gemm_dma_f32_ok(const GemmArgs& a, int batch);
int ff_gemm_dma_f32(const GemmArgs& a, hipStream_t st, int bn = 128);   // bn: tile columns, 128 or 64

__device__ __forceinline__ float ff_sum8(float v) {  // sum over the aligned group of 8 lanes, result on all of them
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  return v;
}

// LDS floats of one ff_gemm_small_tile instance
__host__ __device__ constexpr int ff_gemm_small_lds_floats(int mode, int nw) {
  return nw * 16 * 64 + (mode == 2 ? 32 * 33 : 0) + (mode == 1 ? 64 : 0);
}

// ---- small-M projection: one 32x32 output tile per workgroup, K split over the waves -----------------------------------
template <int KQ, int MODE, int NW>  // MODE: 0 plain, 1 LayerNorm-normalised A rows (+ row table), 2 emits row statistics of C
__device__ __forceinline__ void ff_gemm_small_tile(const GemmArgs& g, int tile, long long bz, float* red) {
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int RPW = 16 / NW;  // accumulator registers (tile rows x 2 halves) a wave finishes
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int m0 = (tile / g.tiles_n) * 32, n0 = (tile % g.tiles_n) * 32;
  const float* Asrc = ((g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A) + bz * g.batch_stride_a;
  int row = m0 + l32, col = n0 + l32;
  row = row < g.M ? row : g.M - 1;
  col = col < g.N ? col : g.N - 1;
  const float* ap = Asrc + (size_t)row * g.lda + wave * KQ + half * 4;
  const float* wp = g.W + bz * g.batch_stride_w + (size_t)col * g.ldw + wave * KQ + half * 4;
  // Loads first: the wave's first operand groups, bias and the epilogue's residual / table values are in flight
  // before anything is waited for; the row statistics (MODE 1) are merged ONCE per block -- wave w takes 32 / NW rows,
  // eight lanes per row, one 16-byte load each -- and handed round through LDS (every wave loading every row's
  // segments itself was half of the kernel's load requests).
#ifndef FF_SMALL_V
#define FF_SMALL_V 2
#endif
  constexpr int V = FF_SMALL_V;            // 1: all operand loads before the first MFMA; 2: groups of 4 k-groups, pipelined
  constexpr int NG = KQ / 8;               // 8-wide k groups per wave
  constexpr int GB = (V == 1) ? (NG < 16 ? NG : 16) : (NG < 4 ? NG : 4);   // groups per batch
  f32x4 a[2][GB], b[2][GB];
#pragma unroll
  for (int j = 0; j < GB; ++j) {
    a[0][j] = ff_ld16(ap + j * 8);
    b[0][j] = ff_ldw16(wp + j * 8);
  }
  float* lnrow = red + NW * 16 * 64 + (MODE == 2 ? 32 * 33 : 0);   // MODE 1: [32][2] (mean, rstd)
  f32x4 sv = {0.f, 0.f, 0.f, 0.f};
  const int spart = lane & 7, srow = wave * (32 / NW) + ((lane >> 3) % (32 / NW));
  if (MODE == 1) {
    int r = m0 + srow;
    r = r < g.M ? r : g.M - 1;
    if (2 * spart < g.ln_nseg) sv = ff_ld16(g.ln_in + ((size_t)r * g.ln_nseg + 2 * spart) * 2);
  }
  const int ocol = n0 + l32;
  const bool colok = ocol < g.N;
  const float bv = (g.bias && colok) ? ff_ldw(g.bias + ocol) : 0.f;
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
    if (g.res) { if (ok) rv[q] = ff_ld4(g.res + bz * g.batch_stride_c + (size_t)orow[q] * g.ldr + ocol); }
    else if (tab && ok && ocol < g.rowtab_cols)
      rv[q] = ff_ldw(g.rowtab + (size_t)(orow[q] / g.rowtab_div) * g.ld_rowtab + ocol);
  }
  __builtin_amdgcn_sched_barrier(0);
  float mu = 0.f, rs = 1.f;
  if (MODE == 1) {
    // Chan's update over the row's 32-column segments, two per lane, eight lanes per row (ln_nseg even, <= 16)
    const bool sok = 2 * spart < g.ln_nseg;
    const float fn = (float)g.ln_nseg;
    const float mean = ff_sum8(sok ? sv.x + sv.z : 0.f) / fn;
    const float m2 = ff_sum8(sok ? sv.y + sv.w : 0.f);
    const float d0 = sv.x - mean, d1 = sv.z - mean;
    const float dev = ff_sum8(sok ? d0 * d0 + d1 * d1 : 0.f);
    const float var = (m2 + 32.f * dev) / (32.f * fn);
    if (spart == 0 && lane < 8 * (32 / NW)) *reinterpret_cast<f32x2*>(lnrow + 2 * srow) = f32x2{mean, 1.0f / sqrtf(var + g.ln_eps)};
    __syncthreads();
    const f32x2 ms = *reinterpret_cast<const f32x2*>(lnrow + 2 * l32);
    mu = ms.x;
    rs = ms.y;
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int g0 = 0; g0 < NG; g0 += GB) {
    const int cur = (g0 / GB) & 1;
    if (g0 + GB < NG) {
#pragma unroll
      for (int j = 0; j < GB; ++j) {
        a[cur ^ 1][j] = ff_ld16(ap + (g0 + GB + j) * 8);
        b[cur ^ 1][j] = ff_ldw16(wp + (g0 + GB + j) * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      f32x4 av = a[cur][j];
      if (MODE == 1) av = (av - mu) * rs;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c], b[cur][j][c], acc, 0, 0, 0);
    }
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
    if (NW == 8)
      v[q] += (red[(4 * 16 + e) * 64 + lane] + red[(5 * 16 + e) * 64 + lane]) +
              (red[(6 * 16 + e) * 64 + lane] + red[(7 * 16 + e) * 64 + lane]);
  }
  float* patch = red + NW * 16 * 64;
  float o[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {   // values first, stores behind them (a store between two uses of loaded operands serialises)
    o[q] = v[q] + bv + (tab ? rv[q] : 0.f);
    if (g.act == 1) o[q] = fmaxf(o[q], 0.f);
    if (!tab) o[q] += rv[q];
    if (MODE == 2) patch[prow[q] * 33 + l32] = o[q];
  }
#pragma unroll
  for (int q = 0; q < RPW; ++q)
    if (colok && orow[q] < g.M) ff_st4(Cout + (size_t)orow[q] * g.ldc + ocol, o[q]);
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
      if (half == 0 && r < g.M) {
        ff_st8(g.ln_out + ((size_t)r * (g.N >> 5) + (n0 >> 5)) * 2, f32x2{mean, m2});
      }
    }
  }
}

// ---- wave-independent attention unit (see ff_attention.hip for the arithmetic) -----------------------------------------
// 2^x on the transcendental unit (v_exp_f32) without the subnormal-range fix-up of exp2f: softmax
// terms below 2^-126 are irrelevant next to a maximum term of 1.
__device__ __forceinline__ float ff_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
constexpr int FF_ATTN_K_LD = 68;   // padded K row of the LDS patches (floats)
__host__ __device__ constexpr int ff_attention_wave_lds_floats(int nwaves) {
  return nwaves * (32 * FF_ATTN_K_LD) + nwaves * 32 + nwaves * (nwaves == 4 ? (3 * 8 * 64 + 40 * 8 + 8) : 0);
}

template <int NWAVES>
__device__ __forceinline__ void ff_attention_wave_block(const ff_attn_desc& d, int q_tiles, int ks, long total_units, int tail_ok,
                                                        int qtail, long vblock, float* lds) {
  constexpr int K_LD = FF_ATTN_K_LD;
  constexpr int PATCH = 32 * K_LD;  // 2176 floats per wave: K tile, later the combine record
  // short-tail mode (four-wave blocks only): rows of up to 8 extra keys (K, V), up to 8 extra queries, their weights
  constexpr int TAILF = NWAVES == 4 ? (3 * 8 * 64 + 40 * 8 + 8) : 0;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;
  float* Kw = lds + wave * PATCH;
  float* Mw = lds + NWAVES * PATCH + wave * 32;
  float* const Kt = lds + NWAVES * (PATCH + 32) + wave * TAILF;   // [8][64] keys 32..39
  float* const Vt = Kt + 512;                                     // [8][64]
  float* const Qt = Kt + 1024;                                    // [8][64] queries 32 * q_tiles ..
  float* const Pw = Kt + 1536;                                    // [40][8] softmax weights of the extra queries
  float* const Mt = Kt + 1856;                                    // [8] additive mask of the extra keys
  const bool tails = NWAVES == 4 && tail_ok != 0;                 // block-uniform

  const int units_per_block = NWAVES / ks;
  const long unit = vblock * units_per_block + wave / ks;
  const int kg = wave % ks;
  const bool unit_valid = unit < total_units;
  const long uc = unit_valid ? unit : total_units - 1;
  const int qt = (int)(uc % q_tiles);
  const int gh_i = (int)(uc / q_tiles);
  const int g = gh_i / d.num_heads, h = gh_i % d.num_heads;

  const int qi = qt * 32 + l32;
  const bool q_valid = unit_valid && qi < d.nq;
  const int qc = qi < d.nq ? qi : d.nq - 1;
  const size_t qrow = (size_t)g * d.q_group_stride + (size_t)(qc / d.q_inner) * d.q_outer_stride +
                      (size_t)(qc % d.q_inner);
  const float qscale = d.scale * 1.4426950408889634f;
  float qreg[32];
  {
    const float* qp = d.q + qrow * d.ldq + h * FF_HEAD_DIM + half * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f32x4 t = ff_ld16(qp + c * 4);
      qreg[c * 4 + 0] = t.x * qscale;
      qreg[c * 4 + 1] = t.y * qscale;
      qreg[c * 4 + 2] = t.z * qscale;
      qreg[c * 4 + 3] = t.w * qscale;
    }
  }
  const float* kbase = d.k + (size_t)g * d.k_group_stride * d.ldk + h * FF_HEAD_DIM;
  const float* vbase = d.v + (size_t)g * d.k_group_stride * d.ldv + h * FF_HEAD_DIM;
  const unsigned char* mrow = d.key_mask ? d.key_mask + (size_t)g * d.mask_stride : nullptr;

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }

  // K staging: lane -> (row = lane/16 + 4p, 16-byte column lane%16): 4 full 256-byte rows per instruction.
  // Addresses are clamped with the STATIC key count (rows up to d.nk exist), the group's own length only masks:
  // the first K and V tiles are requested together with kv_len and the queries -- one round trip, not three.
  const int srow = lane >> 4, sc4 = lane & 15;
  const int nk_s = d.nk;
  const int tiles_s = (nk_s + 31) >> 5;
  f32x4 kst[8];
  float v0[16], v1[16];
  unsigned char mbyte = 0;   // key-mask byte of key (tile, lane), lanes 0..31: travels with the K tile
  auto load_k = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int key = kt * 32 + srow + 4 * p;
      const int kc = key < nk_s ? key : (nk_s > 0 ? nk_s - 1 : 0);
      kst[p] = ff_ld16(kbase + (size_t)kc * d.k_stride * d.ldk + sc4 * 4);
    }
    if (mrow && lane < 32) {
      const int key = kt * 32 + lane;
      mbyte = key < nk_s ? ff_ldw(mrow + key) : (unsigned char)1;
    }
  };
  auto load_v = [&](int kt) {   // V fragments straight to registers (each load instruction reads two full 128-byte row segments)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int kc = key < nk_s ? key : (nk_s > 0 ? nk_s - 1 : 0);
      const float* vp = vbase + (size_t)kc * d.k_stride * d.ldv + l32;
      v0[r] = ff_ld4(vp);
      v1[r] = ff_ld4(vp + 32);
    }
  };
  auto wave_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  int kt = kg;
  if (kt < tiles_s) { load_k(kt); load_v(kt); }
  const bool q_extra = tails && qtail > 0 && unit_valid && qt == q_tiles - 1;   // this wave also serves the extra queries
  f32x4 tk[2], tv[2], tq[2];
  unsigned char tmb = 0;
  if (tails) {   // the extra rows are requested in the same round trip as everything else
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int key = 32 + srow + 4 * p;
      const int kc = key < nk_s ? key : (nk_s > 0 ? nk_s - 1 : 0);
      tk[p] = ff_ld16(kbase + (size_t)kc * d.k_stride * d.ldk + sc4 * 4);
      tv[p] = ff_ld16(vbase + (size_t)kc * d.k_stride * d.ldv + sc4 * 4);
      if (q_extra) {
        const int r = srow + 4 * p;
        const int qx = q_tiles * 32 + (r < qtail ? r : qtail - 1);
        const size_t xr = (size_t)g * d.q_group_stride + (size_t)(qx / d.q_inner) * d.q_outer_stride + (size_t)(qx % d.q_inner);
        tq[p] = ff_ld16(d.q + xr * d.ldq + h * FF_HEAD_DIM + sc4 * 4);
      }
    }
    if (mrow && lane < 8) tmb = (32 + lane) < nk_s ? ff_ldw(mrow + 32 + lane) : (unsigned char)1;
  }
  int nk = nk_s;
  if (d.kv_len) {
    const int kl = ff_ldw(d.kv_len + g);
    nk = kl < nk ? kl : nk;
  }
  if (tails) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      *reinterpret_cast<f32x4*>(Kt + (srow + 4 * p) * 64 + sc4 * 4) = tk[p];
      *reinterpret_cast<f32x4*>(Vt + (srow + 4 * p) * 64 + sc4 * 4) = tv[p];
      if (q_extra) *reinterpret_cast<f32x4*>(Qt + (srow + 4 * p) * 64 + sc4 * 4) = tq[p];
    }
    if (lane < 8) Mt[lane] = ((32 + lane) < nk && tmb == 0) ? 0.f : -INFINITY;
    wave_fence();
  }
  // Short tails (per-sequence self-attention one to eight positions past a multiple of 32: t = 33..37 of the 37- / 38-
  // token configurations) do not get 32-wide MFMA tiles of their own: the extra KEYS are folded into the running
  // softmax on the VALU (lane = query), the extra QUERIES are evaluated after the unit's own tile (lane = key for
  // the scores, lane = head dimension for the values).  Their rows wait in LDS since the first round trip; the main
  // tile's K patch and V fragments are reused -- no further memory request.  36 x 36 scores cost one tile step plus
  // ~3 us instead of four tile steps.
  const int ktail = (tails && nk > 32 && nk <= 40) ? nk - 32 : 0;
  const int ntiles = ktail ? (nk >> 5) : ((nk + 31) >> 5);
  for (; kt < ntiles; kt += ks) {
    // ---- K tile: registers -> private LDS patch -> MFMA fragments ----
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int key = kt * 32 + srow + 4 * p;
      *reinterpret_cast<f32x4*>(Kw + (srow + 4 * p) * K_LD + sc4 * 4) = key < nk ? kst[p] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (mrow) {
      if (lane < 32) {
        const int key = kt * 32 + lane;
        Mw[lane] = (key < nk && mbyte == 0) ? 0.f : -INFINITY;
      }
    }
    wave_fence();
    f32x4 kf[8];
#pragma unroll
    for (int cg = 0; cg < 8; ++cg) kf[cg] = *reinterpret_cast<const f32x4*>(Kw + l32 * K_LD + half * 32 + cg * 4);
    wave_fence();  // fragments are in registers: the K patch may be overwritten (Mw stays valid)
    // (V rows past the group's length are finite values of real rows; their softmax weight is exactly 0)
    if (kt + ks < ntiles) load_k(kt + ks);   // next K tile in flight under the MFMA chains
    // ---- S^T tile ----
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
    for (int cg = 0; cg < 8; ++cg) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg].x, qreg[cg * 4 + 0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg].y, qreg[cg * 4 + 1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg].z, qreg[cg * 4 + 2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg].w, qreg[cg * 4 + 3], s, 0, 0, 0);
    }
    // ---- online softmax ----
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int keyl = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int key = kt * 32 + keyl;
      float mv = mrow ? Mw[keyl] : ((key < nk) ? 0.f : -INFINITY);
      if (d.causal && key > qi) mv = -INFINITY;
      const float v = s[r] + mv;
      s[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = ff_halves_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = ff_exp2(m_run - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = ff_exp2(s[r] - m_safe);
      s[r] = p;
      psum += p;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (!__all(alpha == 1.0f)) {  // the running max moved for some query of this wave
#pragma unroll
      for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0[r], s[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1[r], s[r], o1, 0, 0, 0);
    }
    if (kt + ks < ntiles) load_v(kt + ks);
  }

  if (ktail) {   // (ks == 1, one main tile) keys 32 .. nk-1 for this wave's 32 queries
    float sj[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      sj[jj] = -INFINITY;
      if (jj < ktail) {   // wave-uniform
        const float* kp = Kt + jj * 64 + half * 32;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + c * 4);
          dot += kv.x * qreg[c * 4 + 0] + kv.y * qreg[c * 4 + 1] + kv.z * qreg[c * 4 + 2] + kv.w * qreg[c * 4 + 3];
        }
        dot = ff_halves_sum(dot);
        float sv = dot + Mt[jj];
        if (d.causal && (32 + jj) > qi) sv = -INFINITY;
        sj[jj] = sv;
        tmax = fmaxf(tmax, sv);
      }
    }
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = ff_exp2(m_run - m_safe);
#pragma unroll
    for (int e = 0; e < 16; ++e) { o0[e] *= alpha; o1[e] *= alpha; }
    float psum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      if (jj < ktail) {
        const float pj = ff_exp2(sj[jj] - m_safe);   // 0 for masked keys
        psum += pj;
        const float* vp = Vt + jj * 64 + 4 * half;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(vp + 8 * g4);
          const f32x4 b = *reinterpret_cast<const f32x4*>(vp + 32 + 8 * g4);
          o0[g4 * 4 + 0] += pj * a.x; o0[g4 * 4 + 1] += pj * a.y; o0[g4 * 4 + 2] += pj * a.z; o0[g4 * 4 + 3] += pj * a.w;
          o1[g4 * 4 + 0] += pj * b.x; o1[g4 * 4 + 1] += pj * b.y; o1[g4 * 4 + 2] += pj * b.z; o1[g4 * 4 + 3] += pj * b.w;
        }
      }
    }
    l_run = l_run * alpha + (half == 0 ? psum : 0.f);   // both halves hold the same weights: count them once
    m_run = m_new;
  }

  float* const op = d.o + qrow * d.ldo + h * FF_HEAD_DIM + 4 * half;
  if (ks > 1) {
    // ---- combine the ks key groups of a unit through LDS (record: O[32 regs][64 lanes], m, l).  Every wave of the
    //      unit takes 32 / ks of the output registers and sums them over the records in ascending key-group order.
    wave_fence();
#pragma unroll
    for (int e = 0; e < 16; ++e) { Kw[e * 64 + lane] = o0[e]; Kw[(16 + e) * 64 + lane] = o1[e]; }
    Kw[32 * 64 + lane] = m_run;   // 2048 + 64 + 64 = 2176 = PATCH exactly
    Kw[33 * 64 + lane] = l_run;
    __syncthreads();
    const float* rec0 = lds + (wave - kg) * PATCH;
    float sc[8];
    float m_star = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < ks) { sc[j] = rec0[j * PATCH + 32 * 64 + lane]; m_star = fmaxf(m_star, sc[j]); }
    const float ms = (m_star == -INFINITY) ? 0.f : m_star;
    float l_sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < ks) { sc[j] = ff_exp2(sc[j] - ms); l_sum += rec0[j * PATCH + 33 * 64 + lane] * sc[j]; }
    const float l_all = ff_halves_sum(l_sum);
    const float inv_c = l_all > 0.f ? 1.0f / l_all : 0.f;
    const int quads = q_valid ? 8 / ks : 0;   // groups of four output registers per wave (ks = 2, 4, 8)
    for (int qd = 0; qd < quads; ++qd) {
      const int e0 = (kg * quads + qd) * 4;   // registers e0 .. e0+3: o0 for e0 < 16, else o1
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < ks) {
          const float* r = rec0 + j * PATCH + e0 * 64 + lane;
          acc.x += r[0] * sc[j];
          acc.y += r[64] * sc[j];
          acc.z += r[128] * sc[j];
          acc.w += r[192] * sc[j];
        }
      acc.x *= inv_c; acc.y *= inv_c; acc.z *= inv_c; acc.w *= inv_c;
      ff_st16(op + (e0 < 16 ? 2 * e0 : 32 + 2 * (e0 - 16)), acc);
    }
    return;   // (the key-split form never carries short tails: nothing below applies)
  }

  const float l_tot = ff_halves_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_valid) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 a = {o0[g4 * 4 + 0] * inv, o0[g4 * 4 + 1] * inv, o0[g4 * 4 + 2] * inv, o0[g4 * 4 + 3] * inv};
      f32x4 b = {o1[g4 * 4 + 0] * inv, o1[g4 * 4 + 1] * inv, o1[g4 * 4 + 2] * inv, o1[g4 * 4 + 3] * inv};
      ff_st16(op + 8 * g4, a);
      ff_st16(op + 32 + 8 * g4, b);
    }
  }

  if (q_extra) {   // (ks == 1, nk <= 40) queries 32 * q_tiles .. nq-1 of this (group, head)
    // scores: lane = key; K rows 0..31 are still in the patch (one main tile), rows 32.. in Kt
    const int lk = lane < 40 ? lane : 39;
    const float* krp = lane < 32 ? Kw + lane * K_LD : Kt + (lk - 32) * 64;
    f32x4 kr[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) kr[c] = *reinterpret_cast<const f32x4*>(krp + c * 4);
    bool kok = lane < nk;
    if (mrow && kok) kok = (lane < 32 ? Mw[lane] : Mt[lk - 32]) == 0.f;
    f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < qtail) {   // wave-uniform
        const float* qp = Qt + i * 64;
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const f32x4 qv = *reinterpret_cast<const f32x4*>(qp + c * 4);
          dot += kr[c].x * qv.x + kr[c].y * qv.y + kr[c].z * qv.z + kr[c].w * qv.w;
        }
        const bool ok = kok && !(d.causal && lane > q_tiles * 32 + i);
        const float sv = ok ? dot * qscale : -INFINITY;
        const float mx = ff_wave_max(sv);
        const float pe = ok ? ff_exp2(sv - mx) : 0.f;
        const float ls = ff_wave_sum(pe);
        const float pn = ls > 0.f ? pe / ls : 0.f;   // normalised weight of (query i, key lane)
        if (i < 4) pa[i & 3] = pn; else pb[i & 3] = pn;
      }
    }
    if (lane < 40) {
      *reinterpret_cast<f32x4*>(Pw + lane * 8) = pa;
      *reinterpret_cast<f32x4*>(Pw + lane * 8 + 4) = pb;
    }
    wave_fence();
    // values: lane = (head dimension l32 | 32 + l32, key half); the main tile's V fragments are still in v0 / v1
    float a0[8], a1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
      const f32x4 wa = *reinterpret_cast<const f32x4*>(Pw + key * 8);
      const f32x4 wb = *reinterpret_cast<const f32x4*>(Pw + key * 8 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] += wa[i] * v0[r]; a1[i] += wa[i] * v1[r];
        a0[4 + i] += wb[i] * v0[r]; a1[4 + i] += wb[i] * v1[r];
      }
    }
    const float once = half == 0 ? 1.f : 0.f;   // the extra keys are not split between the halves
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      if (jj < ktail) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(Pw + (32 + jj) * 8);
        const f32x4 wb = *reinterpret_cast<const f32x4*>(Pw + (32 + jj) * 8 + 4);
        const float x0 = Vt[jj * 64 + l32] * once, x1 = Vt[jj * 64 + 32 + l32] * once;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a0[i] += wa[i] * x0; a1[i] += wa[i] * x1;
          a0[4 + i] += wb[i] * x0; a1[4 + i] += wb[i] * x1;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < qtail) {
        const float s0 = ff_halves_sum(a0[i]);
        const float s1 = ff_halves_sum(a1[i]);
        const int qx = q_tiles * 32 + i;
        const size_t xr = (size_t)g * d.q_group_stride + (size_t)(qx / d.q_inner) * d.q_outer_stride + (size_t)(qx % d.q_inner);
        ff_st4(d.o + xr * d.ldo + h * FF_HEAD_DIM + 32 * half + l32, half ? s1 : s0);
      }
    }
  }
}

// ---- LayerNorm (+pos) of one row by one wavefront ---------------------------------------------------------------------
// NV = float4 chunks per lane (E <= 256*NV).  Two-pass statistics in registers (mean, then the centred second moment) --
// the same formula torch's CPU kernel evaluates, biased variance.
struct LnArgs {
  const float* x; int ldx;
  const float* gamma; const float* beta; float eps;
  float* y; int ldy;
  float* ypos; int ldypos;
  const float* pos; int ldpos, pos_div, pos_mod;
  int rows, E;
};

template <int NV>
__device__ __forceinline__ void ff_layernorm_row(const LnArgs& a, int row, int lane) {
  const int nvec = a.E >> 2;
  const float* xr = a.x + (size_t)row * a.ldx;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    int vi = lane + c * 64;
    if (vi < nvec) {
      v[c] = ff_ld16(xr + vi * 4);
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
    } else {
      v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float inv_e = 1.0f / (float)a.E;
  const float mean = ff_wave_sum(s) * inv_e;
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    int vi = lane + c * 64;
    if (vi < nvec) {
      f32x4 d = v[c] - mean;
      ss += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  }
  const float var = ff_wave_sum(ss) * inv_e;
  const float rstd = 1.0f / sqrtf(var + a.eps);
  const float* pr = nullptr;
  if (a.ypos != nullptr) pr = a.pos + (size_t)((row / a.pos_div) % a.pos_mod) * a.ldpos;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    int vi = lane + c * 64;
    if (vi < nvec) {
      f32x4 g = ff_ldw16(a.gamma + vi * 4);
      f32x4 b = ff_ldw16(a.beta + vi * 4);
      f32x4 o = (v[c] - mean) * rstd * g + b;
      if (a.y != nullptr) ff_st16(a.y + (size_t)row * a.ldy + vi * 4, o);
      if (a.ypos != nullptr) {
        f32x4 p = ff_ldw16(pr + vi * 4);
        ff_st16(a.ypos + (size_t)row * a.ldypos + vi * 4, o + p);
      }
    }
  }
}

// ---- pointer head, stage 2: mask + (value, index) reduction + feedback gather of one sequence by one wavefront ---------
struct PointerArgs {
  const float* p; int ldp;
  const float* memory; int S, E;
  const unsigned char* mask; const int* kv_len;
  const unsigned char* extra; int ldextra;
  int B, spg;
  int* next_tok; float* best; float* second; float* logits; int ldlogits;
  float* next_rows; int ldnext;
  int* count_ge; int ge_bound; int* count_eq; int eq_value;
  // Decode engine only (null otherwise): stop-rule counter of THIS launch published to the host without a copy launch.
  int* seen;        // [B] seq2seq, FF_STOP_EACH_EOS: count_eq counts a sequence's FIRST eq_value only
  int* arrive;      // arrivals of this launch's B sequences (zeroed before the decode)
  int* host_slot;   // host-mapped pinned int: the last sequence to arrive stores the launch's counter there (system scope)
  int host_which;   // 0: count_ge, 1: count_eq
  float* next_stats;  // [B, E/32, 2] or null: LayerNorm segment statistics of the appended rows (E % 32 == 0)
};

// Stop-rule counters of the (up to four) sequences a 256-thread block has finished, by ONE thread of the block: one atomic per
// counter and block instead of one per sequence (4096 sequences of a 16-wireframe micro-batch adding to one address, plus the
// arrival count, had taken pointer_reduce_kernel from 54 to 119 us), and -- when the engine asks for it -- the launch's total
// into host-mapped memory by the LAST block to get here: every block's adds are ordered before its arrival (release), the last
// arrival reads the counter after it (acquire).  The host looks at the slot only behind an event recorded after the launch.
// toks[i]: token of sequence b0 + i, nvalid of them exist.
__device__ __forceinline__ void ff_pointer_count_block(const PointerArgs& a, int b0, const int* toks, int nvalid) {
  int nge = 0, neq = 0;
  for (int i = 0; i < nvalid; ++i) {
    const int idx = toks[i];
    if (a.count_ge && idx >= a.ge_bound) ++nge;
    if (a.count_eq && idx == a.eq_value) {
      bool first = true;
      if (a.seen) { first = a.seen[b0 + i] == 0; a.seen[b0 + i] = 1; }   // (only this sequence's block touches seen[b], one step at a time)
      if (first) ++neq;
    }
  }
  if (nge) atomicAdd(a.count_ge, nge);
  if (neq) atomicAdd(a.count_eq, neq);
  if (a.arrive) {
    const int prev = __hip_atomic_fetch_add(a.arrive, nvalid, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + nvalid == a.B) {
      const int v = __hip_atomic_load(a.host_which ? a.count_eq : a.count_ge, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.host_slot, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// `logits` holds the raw dot products of every (sequence, key); mask the row in place and reduce (value, index) pairs --
// per lane over its strided keys, then across the 64 lanes with a butterfly that keeps torch's tie rule (lowest index)
// and the runner-up value.
// Returns the selected token (wave-uniform); the caller counts it (ff_pointer_count_block).
__device__ __forceinline__ int ff_pointer_reduce_row(const PointerArgs& a, int b, int lane) {
  const int w = b / a.spg;
  int kv = a.S;
  if (a.kv_len) { const int k = ff_ldw(a.kv_len + w); kv = k < kv ? k : kv; }
  const unsigned char* mrow = a.mask ? a.mask + (size_t)w * a.S : nullptr;
  const unsigned char* erow = a.extra ? a.extra + (size_t)b * a.ldextra : nullptr;
  float* lrow = a.logits + (size_t)b * a.ldlogits;
  const float FILL = -3.402823466e+38f;  // -FLT_MAX = torch.finfo(float32).min (reference utils.py:16-20)
  float b1 = -INFINITY, b2 = -INFINITY;
  int i1 = 0x7fffffff;
  for (int s = lane; s < a.S; s += 64) {
    bool ok = s < kv;
    if (ok && mrow) ok = ff_ldw(mrow + s) == 0;
    if (ok && erow) ok = ff_ldw(erow + s) == 0;
    const float v = ok ? ff_ld4(lrow + s) : FILL;
    ff_st4(lrow + s, v);
    if (v > b1) { b2 = b1; b1 = v; i1 = s; }
    else if (v > b2) b2 = v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob1 = __shfl_xor(b1, off, FF_WAVE), ob2 = __shfl_xor(b2, off, FF_WAVE);
    const int oi1 = __shfl_xor(i1, off, FF_WAVE);
    const bool other = (ob1 > b1) || (ob1 == b1 && oi1 < i1);
    const float nb2 = other ? fmaxf(ob2, b1) : fmaxf(b2, ob1);
    if (other) { b1 = ob1; i1 = oi1; }
    b2 = nb2;
  }
  if (i1 == 0x7fffffff) { i1 = 0; b1 = FILL; }
  if (lane == 0) {
    ff_st4i(a.next_tok + b, i1);
    if (a.best) ff_st4(a.best + b, b1);
    if (a.second) ff_st4(a.second + b, b2);
  }
  if (a.next_rows) {
    const float* src = a.memory + ((size_t)w * a.S + i1) * a.E;
    float* dst = a.next_rows + (size_t)b * a.ldnext;
    if (!a.next_stats) {
      for (int vi = lane; vi < (a.E >> 2); vi += 64)
        ff_st16(dst + vi * 4, ff_ldw16(src + vi * 4));
    } else {
      // the same copy, leaving (mean, M2) of every 32-column segment of the row (two passes like the LayerNorm kernel and the
      // statistics-producing GEMM epilogue): a segment is the 8 float4 of 8 consecutive lanes
      float* sdst = a.next_stats + (size_t)b * (a.E >> 5) * 2;
      for (int v0 = 0; v0 < (a.E >> 2); v0 += 64) {      // (wave-uniform bound: the shuffles below need every lane)
        const int vi = v0 + lane;
        const bool in = vi < (a.E >> 2);
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (in) { x = ff_ldw16(src + vi * 4); ff_st16(dst + vi * 4, x); }
        float sm = (x.x + x.y) + (x.z + x.w);
        sm += __shfl_xor(sm, 1, FF_WAVE); sm += __shfl_xor(sm, 2, FF_WAVE); sm += __shfl_xor(sm, 4, FF_WAVE);
        const float mean = sm * (1.0f / 32.0f);
        const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
        float m2 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        m2 += __shfl_xor(m2, 1, FF_WAVE); m2 += __shfl_xor(m2, 2, FF_WAVE); m2 += __shfl_xor(m2, 4, FF_WAVE);
        if (in && (lane & 7) == 0) ff_st8(sdst + (vi >> 3) * 2, f32x2{mean, m2});
      }
    }
  }
  return i1;
}
