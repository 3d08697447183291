// Multi-head attention core for the decode path (encoder self-, decoder self-, decoder cross-
// attention) on the f32 matrix cores, flash style.
//
// One wavefront owns 32 queries of one (group, head); a block is NW such waves sharing the group's
// key/value rows, which are staged through LDS in chunks of 64 keys (coalesced 16-byte loads; K rows
// padded to 68 floats so the per-lane ds_read_b128 of a key row is bank-conflict free).
//
// Both products keep the QUERY index on the lane (l & 31) so the running max / sum / rescale of the
// online softmax never leave the lane:
//   S^T[key][q] = sum_d K[key][d] * Q[q][d]    A = K (LDS), B = Q (registers, pre-scaled)
//   O^T[d][q]   = sum_key V[key][d] * P[q][key] A = V (LDS), B = P = exp(S^T - m) (the S accumulator
//                                               registers themselves: the C-layout key index of
//                                               register r, (r&3) + 8*(r>>2) + 4*(l>>5), is used as
//                                               the k index of MFMA step r for A and B alike)
// so P never moves between lanes or through LDS.  The two lane halves hold disjoint key subsets of
// a query; they exchange only the tile max (one __shfl_xor 32) and add their partial sums at the end.
#include <math.h>

#include <atomic>

#include "ff_common.h"
#include "ff_device.h"

// Timing experiment (tools/attn_phase_probe.py, -DFF_EXP_ATTN_STAMP): workgroup 0, wave 0 of the K/V-resident kernel stamps the
// shader clock at entry, when K / V are in LDS, when its items are done, after the partial records are exchanged, at the end.
#ifdef FF_EXP_ATTN_STAMP
__device__ unsigned long long ff_exp_attn_stamps[8 + 16];
extern "C" int ff_exp_read_attn_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_exp_attn_stamps), sizeof(ff_exp_attn_stamps)) == hipSuccess ? 0 : -1;
}
#define FF_EXP_ASTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) ff_exp_attn_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define FF_EXP_ASTAMP(i) do { } while (0)
#endif

namespace {


constexpr int KC = 64;     // keys per LDS chunk
constexpr int K_LD = FF_ATTN_K_LD;   // padded K row (floats)
constexpr int V_LD = 64;

template <int NW, bool DB>
__global__ __launch_bounds__(64 * NW) void attention_kernel(ff_attn_desc d, int q_tiles) {
  constexpr int NBUF = DB ? 2 : 1;
  constexpr int CHUNK_FLOATS = KC * K_LD + KC * V_LD + KC;
  __shared__ __attribute__((aligned(16))) float lds[NBUF * CHUNK_FLOATS];

  constexpr int NT = 64 * NW;
  constexpr int NLD = (KC * 16) / NT;  // float4 loads per thread per operand per chunk
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int half = lane >> 5, l32 = lane & 31;

  // block -> (query tile, group, head); (group, head) varies fastest so that a (group, head) pair
  // keeps landing on the same XCD (block b -> XCD b % 8) and its K/V stay in that L2.
  const int gh = d.num_groups * d.num_heads;
  const int qt = blockIdx.x / gh;
  const int rem = blockIdx.x % gh;
  const int g = rem / d.num_heads, h = rem % d.num_heads;

  const int qi = (qt * NW + wave) * 32 + l32;
  const bool q_valid = qi < d.nq;
  const int qc = q_valid ? qi : d.nq - 1;
  const size_t qrow = (size_t)g * d.q_group_stride + (size_t)(qc / d.q_inner) * d.q_outer_stride +
                      (size_t)(qc % d.q_inner);

  // scores are kept in the log2 domain: q is pre-multiplied by scale * log2(e) so that the softmax
  // numerator is a bare v_exp_f32 (2^x); the result is the same softmax.
  const float qscale = d.scale * 1.4426950408889634f;
  float qreg[32];
  {
    const float* qp = d.q + qrow * d.ldq + h * FF_HEAD_DIM + half * 32;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f32x4 t = *reinterpret_cast<const f32x4*>(qp + c * 4);
      qreg[c * 4 + 0] = t.x * qscale;
      qreg[c * 4 + 1] = t.y * qscale;
      qreg[c * 4 + 2] = t.z * qscale;
      qreg[c * 4 + 3] = t.w * qscale;
    }
  }

  int nk = d.nk;
  if (d.kv_len) {
    int kl = d.kv_len[g];
    nk = kl < nk ? kl : nk;
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o0[e] = 0.f; o1[e] = 0.f; }

  const float* kbase = d.k + (size_t)g * d.k_group_stride * d.ldk + h * FF_HEAD_DIM;
  const float* vbase = d.v + (size_t)g * d.k_group_stride * d.ldv + h * FF_HEAD_DIM;

  // ---- staging: global -> registers -> LDS.  Only the rows of the 32-key tiles that will be
  //      computed are touched (rows past nk inside such a tile are zero-filled: P is 0 there, but
  //      0 * garbage must not produce NaN). ----
  constexpr int NREG = DB ? NLD : 1;  // the single-buffered variant stages in small batches
  f32x4 kreg[NREG], vreg[NREG];
  float mreg = 0.f;
  auto rows_needed = [&](int c0) { int r = nk - c0; r = r > KC ? KC : r; return (r + 31) & ~31; };
  auto load_chunk = [&](int c0) {
    const int need = rows_needed(c0);
#pragma unroll
    for (int p = 0; p < NREG; ++p) {
      const int idx = tid + p * NT;
      const int row = idx >> 4, c4 = idx & 15;
      const int key = c0 + row;
      kreg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      vreg[p] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < need && key < nk) {
        const size_t krow = (size_t)key * d.k_stride;
        kreg[p] = *reinterpret_cast<const f32x4*>(kbase + krow * d.ldk + c4 * 4);
        vreg[p] = *reinterpret_cast<const f32x4*>(vbase + krow * d.ldv + c4 * 4);
      }
    }
    if (tid < KC) {
      const int key = c0 + tid;
      bool masked = key >= nk;
      if (!masked && d.key_mask) masked = d.key_mask[(size_t)g * d.mask_stride + key] != 0;
      mreg = masked ? -INFINITY : 0.f;
    }
  };
  auto store_chunk = [&](int c0, int buf) {
    const int need = rows_needed(c0);
    float* Ks = lds + buf * CHUNK_FLOATS;
    float* Vs = Ks + KC * K_LD;
    float* Ms = Vs + KC * V_LD;
#pragma unroll
    for (int p = 0; p < NREG; ++p) {
      const int idx = tid + p * NT;
      const int row = idx >> 4, c4 = idx & 15;
      if (row < need) {
        *reinterpret_cast<f32x4*>(Ks + row * K_LD + c4 * 4) = kreg[p];
        *reinterpret_cast<f32x4*>(Vs + row * V_LD + c4 * 4) = vreg[p];
      }
    }
    if (tid < KC) Ms[tid] = mreg;
  };

  auto compute_chunk = [&](int c0, int buf) {
    const float* Ks = lds + buf * CHUNK_FLOATS;
    const float* Vs = Ks + KC * K_LD;
    const float* Ms = Vs + KC * V_LD;
#pragma unroll
    for (int kt = 0; kt < KC / 32; ++kt) {
      if (c0 + kt * 32 >= nk) break;  // block-uniform
      // ---- S^T tile: 32 keys x 32 queries ----
      f32x16 s;
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = 0.f;
      const float* krow_p = Ks + (kt * 32 + l32) * K_LD + half * 32;
#pragma unroll
      for (int cg = 0; cg < 8; ++cg) {
        f32x4 kf = *reinterpret_cast<const f32x4*>(krow_p + cg * 4);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[cg * 4 + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[cg * 4 + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[cg * 4 + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[cg * 4 + 3], s, 0, 0, 0);
      }
      // ---- mask, online softmax (query on the lane; keys across registers and the two halves) ----
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int keyl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = s[r] + Ms[keyl];
        if (d.causal && (c0 + keyl) > qi) v = -INFINITY;
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
        const int keyl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v0 = Vs[keyl * V_LD + l32];
        const float v1 = Vs[keyl * V_LD + 32 + l32];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[r], o1, 0, 0, 0);
      }
    }
  };

  if (DB) {
    if (nk > 0) {
      load_chunk(0);
      store_chunk(0, 0);
    }
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < nk; c0 += KC) {
      const bool more = (c0 + KC) < nk;
      if (more) load_chunk(c0 + KC);
      compute_chunk(c0, buf);
      if (more) store_chunk(c0 + KC, buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  } else {
    for (int c0 = 0; c0 < nk; c0 += KC) {
      __syncthreads();
      {
        const int need = rows_needed(c0);
        float* Ks = lds;
        float* Vs = Ks + KC * K_LD;
        float* Ms = Vs + KC * V_LD;
#pragma unroll 4
        for (int p = 0; p < NLD; ++p) {
          const int idx = tid + p * NT;
          const int row = idx >> 4, c4 = idx & 15;
          const int key = c0 + row;
          if (row < need) {
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < nk) {
              const size_t krow = (size_t)key * d.k_stride;
              kv = *reinterpret_cast<const f32x4*>(kbase + krow * d.ldk + c4 * 4);
              vv = *reinterpret_cast<const f32x4*>(vbase + krow * d.ldv + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(Ks + row * K_LD + c4 * 4) = kv;
            *reinterpret_cast<f32x4*>(Vs + row * V_LD + c4 * 4) = vv;
          }
        }
        if (tid < KC) {
          const int key = c0 + tid;
          bool masked = key >= nk;
          if (!masked && d.key_mask) masked = d.key_mask[(size_t)g * d.mask_stride + key] != 0;
          Ms[tid] = masked ? -INFINITY : 0.f;
        }
      }
      __syncthreads();
      compute_chunk(c0, 0);
    }
  }

  const float l_tot = ff_halves_sum(l_run);
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  if (q_valid) {
    float* op = d.o + qrow * d.ldo + h * FF_HEAD_DIM + 4 * half;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 a = {o0[g4 * 4 + 0] * inv, o0[g4 * 4 + 1] * inv, o0[g4 * 4 + 2] * inv, o0[g4 * 4 + 3] * inv};
      f32x4 b = {o1[g4 * 4 + 0] * inv, o1[g4 * 4 + 1] * inv, o1[g4 * 4 + 2] * inv, o1[g4 * 4 + 3] * inv};
      ff_st16(op + 8 * g4, a);
      ff_st16(op + 32 + 8 * g4, b);
    }
  }
}

// ---- wave-independent variant ------------------------------------------------------------------------
// No block barrier and no shared staging: every wave owns one unit = (32 queries of one (group, head)) x
// (every ks-th 32-key tile).  K tiles go through a wave-private LDS patch (coalesced row loads in,
// MFMA-fragment ds_read_b128 out; LDS operations of one wave are ordered, so only a compiler fence
// sits between them), V fragments are loaded straight into registers (each load instruction reads two
// full 128-byte row segments), the next K tile is prefetched into registers under the MFMA chains.
// With ks > 1 the ks waves that share the queries combine their (max, sum, O) through LDS at the end --
// the launch then has ks x more independent units, which is what the small-t steps of the decode
// need (a (wireframe, head) pair offers only F*t/32 query tiles; at t = 1 that is 64 units on 1024 SIMDs).
template <int NWAVES>
__global__ __launch_bounds__(64 * NWAVES, (NWAVES == 4 ? 2 : 1)) void attention_wave_kernel(ff_attn_desc d, int q_tiles, int ks,
                                                                      long total_units, int tail_ok, int qtail) {
  __shared__ __attribute__((aligned(16))) float lds[ff_attention_wave_lds_floats(NWAVES)];
  ff_attention_wave_block<NWAVES>(d, q_tiles, ks, total_units, tail_ok, qtail, (long)blockIdx.x, lds);
}

// ---- K/V-resident variant ------------------------------------------------------------------------------
// For key sets of at most 288 rows (every decoder cross-attention of the 64..256-edge configurations, and the
// encoder): the K and V rows of ONE (group, head) pair -- <= 2 x 72 KB -- are loaded into LDS once per block and
// stay there; the pair's query tiles are dealt out to `c` blocks (one per CU), so the K/V bytes leave L2 once per
// block instead of once per 128 (block-shared kernel) or 32 (wave kernel) queries.
// Work balance: a block's work is the list of (query tile, key tile) ITEMS of its query tiles, cut into eight equal
// consecutive ranges, one per wave (two per SIMD).  Whole query tiles inside a range are finished by their wave; a
// query tile cut between waves leaves partial (max, sum, O) records, which meet in LDS (the K/V area is free by
// then) and are merged in ascending key order by the wave that holds the tile's first keys: deterministic, and the
// per-SIMD load differs by at most one key tile -- no "2304 tiles on 2048 slots take two rounds" cliff.
constexpr int RK_KEYS = 288;                     // 9 key tiles
constexpr int RK_NW = 8;
constexpr int RK_REC = 34 * 64;                  // one partial record: O[32 regs][64 lanes], m[64], l[64]
constexpr int RK_LDS_FLOATS = RK_KEYS * 128 + RK_KEYS;
static_assert(RK_NW * RK_REC <= RK_KEYS * 128, "records must fit the K/V area");

// V operand reads of the K/V-resident kernel, issued by hand (see the P.V loop): rows r = 2 PR, 2 PR + 1 of the accumulator layout
// ((r & 3) + 8 (r >> 2) key rows of 64 floats past `va`), columns l32 and 32 + l32 -> dst[0..3].
template <int PR>
__device__ __forceinline__ void rk_read_v(float (&dst)[4], unsigned va) {
  constexpr int R0 = 2 * PR, R1 = 2 * PR + 1;
  constexpr int O0 = ((R0 & 3) + 8 * (R0 >> 2)) * 256, O1 = ((R1 & 3) + 8 * (R1 >> 2)) * 256;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[0]) : "v"(va), "n"(O0) : "memory");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[1]) : "v"(va), "n"(O0 + 128) : "memory");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[2]) : "v"(va), "n"(O1) : "memory");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst[3]) : "v"(va), "n"(O1 + 128) : "memory");
}
__device__ __forceinline__ void rk_read_v_dyn(float (&dst)[4], unsigned va, int pr) {   // pr is a compile-time constant after unrolling
  switch (pr) {
    case 1: rk_read_v<1>(dst, va); break;
    case 2: rk_read_v<2>(dst, va); break;
    case 3: rk_read_v<3>(dst, va); break;
    case 4: rk_read_v<4>(dst, va); break;
    case 5: rk_read_v<5>(dst, va); break;
    case 6: rk_read_v<6>(dst, va); break;
    default: rk_read_v<7>(dst, va); break;
  }
}

struct RkState {
  float m, l;
  f32x16 o0, o1;
};

__global__ __launch_bounds__(64 * RK_NW, 2) void attention_resident_kernel(ff_attn_desc d, int P, int c, int q_tiles, int w_old, int w_young, int phase_knob,
                                                                          float* __restrict__ scratch, int rot_on) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Ks = lds;                       // [key][64], 16-byte chunks XOR-swizzled with (key & 15)
  float* const Vs = lds + RK_KEYS * 64;        // [key][64]
  float* const Ms = lds + RK_KEYS * 128;       // additive key bias: 0 or -inf
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const float qscale = d.scale * 1.4426950408889634f;
  FF_EXP_ASTAMP(0);

  const int nblk = gridDim.x;
  const int rank = (P <= nblk) ? blockIdx.x / P : 0;           // which share of the pair's query tiles
  for (int pair = (P <= nblk) ? blockIdx.x % P : blockIdx.x; pair < P; pair += (P <= nblk ? P : nblk)) {
    const int g = pair / d.num_heads, h = pair % d.num_heads;
    const float* kbase = d.k + (size_t)g * d.k_group_stride * d.ldk + h * FF_HEAD_DIM;
    const float* vbase = d.v + (size_t)g * d.k_group_stride * d.ldv + h * FF_HEAD_DIM;
    // ---- everything the block needs from memory is requested in ONE round trip: kv_len, the mask bytes, K and V
    //      (LDS-DMA: no staging registers, all 2 x 9 tiles in flight), the first query tile of every wave ----
    const int tiles_max = (d.nk + 31) >> 5;               // static bound; the group's own length arrives meanwhile
    int nk = d.nk;
    if (d.kv_len) { const int kl = d.kv_len[g]; nk = kl < nk ? kl : nk; }
    unsigned char mbyte = 0;
    if (d.key_mask && tid < tiles_max * 32 && tid < d.nk) mbyte = d.key_mask[(size_t)g * d.mask_stride + tid];
    {
      // piece = 4 key rows x 256 B of one operand = one wave-instruction (lane: row lane/16, 16-byte slot lane%16);
      // the LDS image is lane-linear, so K's bank swizzle (slot ^ (row & 15)) is applied to the SOURCE column.
      // Rows past d.nk read the last row again (finite values; their softmax weight is exactly 0).
      // The c blocks of a pair sit on the same XCD and read the SAME rows: started together on the same piece they queue on
      // the same L2 channels, so block `rank` starts rot = rank / c of the way round (destination = piece id, only the order moves).
      const int npieces = tiles_max * 8;
      const int prow = lane >> 4, pos = lane & 15;
      const int rot = (P <= nblk && rot_on) ? (int)(((long)rank * npieces) / c) : 0;
      for (int q = wave; q < npieces; q += RK_NW) {
        const int p = q + rot < npieces ? q + rot : q + rot - npieces;
        const int row = 4 * p + prow;
        const int rc = row < d.nk ? row : d.nk - 1;
        const size_t krow = (size_t)rc * d.k_stride;
        __builtin_amdgcn_global_load_lds(kbase + krow * d.ldk + ((pos ^ (row & 15)) << 2),
                                         (__attribute__((address_space(3))) void*)(Ks + p * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(vbase + krow * d.ldv + (pos << 2),
                                         (__attribute__((address_space(3))) void*)(Vs + p * 256), 16, 0, 0);
      }
    }
    const int ntiles_all = (nk + 31) >> 5;
    // A last key tile with at most 4 keys (S = edges + 4 special tokens with a multiple of 32 edges: every
    // benchmark configuration) is not worth a 32x32 MFMA tile (64 instructions for <= 1/4 useful work): its scores
    // and its P.V contribution are evaluated on the VALU (LDS broadcast reads) as an appendix of the last full tile.
    const int tail = nk - (ntiles_all - 1) * 32;
    const bool vtail = ntiles_all > 1 && tail <= 4;
    const int ntiles = vtail ? ntiles_all - 1 : ntiles_all;   // MFMA key tiles = work items per query tile
    const int nq_blk = rank < q_tiles ? (q_tiles - rank + c - 1) / c : 0;
    // ---- this wave's item range ----
    // Uneven shares: a SIMD issues its older wave (0..3) first, the younger one (4..7) fills the gaps and then runs its last
    // items alone (phase probe, profiles/r03/attn_phase_probe.txt: 76 k vs 101-107 k cycles with equal shares), so the older
    // waves take w_old parts and the younger ones w_young parts of every SIMD's items.
    const int I = nq_blk * ntiles;
    const int wsum = 4 * (w_old + w_young);
    // range_of(w): first and one-past-last item of wave w (ascending with the wave index: the merge below relies on it)
    auto range_of = [&](int w, int& a, int& b) {
      const int pre0 = w < 4 ? w * w_old : 4 * w_old + (w - 4) * w_young;
      const int pre1 = pre0 + (w < 4 ? w_old : w_young);
      a = (int)(((long)pre0 * I) / wsum); b = (int)(((long)pre1 * I) / wsum);
    };
    int ia, ib;
    range_of(wave, ia, ib);
    auto query_row = [&](int qt, bool& valid) -> size_t {
      const int qi = qt * 32 + l32;
      valid = qi < d.nq;
      const int qc = valid ? qi : d.nq - 1;
      return (size_t)g * d.q_group_stride + (size_t)(qc / d.q_inner) * d.q_outer_stride + (size_t)(qc % d.q_inner);
    };
    f32x4 qraw[8];   // query rows of the wave's current tile (one register set: a second one for prefetching every
                     // tile spills -- 256 VGPRs + scratch; the first tile is fetched under the K/V transfer)
    auto fetch_q = [&](int qt) {
      bool qv;
      const float* qp = d.q + query_row(qt, qv) * d.ldq + h * FF_HEAD_DIM + half * 32;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) qraw[cc] = *reinterpret_cast<const f32x4*>(qp + cc * 4);
    };
    if (ia < ib) fetch_q(rank + (ia / ntiles) * c);
    if (tid < tiles_max * 32) Ms[tid] = (tid >= nk || mbyte != 0) ? -INFINITY : 0.f;
    __syncthreads();   // (waits for the LDS-DMA: it counts in vmcnt)
    FF_EXP_ASTAMP(1);
    // probe knobs (FF_RK_PHASE): de-phase the two waves of a SIMD -- 1: the younger waves (4..7) at priority 1, 2: they start
    // ~1500 cycles late, 3: both, 4: the OLDER waves at priority 1
    if ((phase_knob & 1) && phase_knob < 4 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    if (phase_knob == 4 && wave < 4) __builtin_amdgcn_s_setprio(1);
    if ((phase_knob & 2) && phase_knob < 4 && wave >= 4) __builtin_amdgcn_s_sleep(23);

    // ---- one query tile x key tiles [kt0, kt1), software-pipelined inside the wave ----
    // The matrix pipe is the unit to keep busy: an item is 33 + 32 MFMAs (64 cycles each) and ~600 cycles of softmax VALU work
    // between them, and two waves per SIMD that alternate MFMAs fall into lock step -- both in the softmax at the same time,
    // the pipe idle (phase probe, round 3: 9.9 k cycles per pair of items for 8.2 k of MFMA time, and the younger wave of a SIMD
    // finishes 11-25 k cycles after the older one).  So the wave itself overlaps them: while the P.V products of key tile kt
    // issue, the VALU turns the scores of tile kt+1 (computed just before) into weights.  Two score register sets (sa / sb)
    // alternate roles; the arithmetic and its order per query are unchanged (rescale by alpha(kt+1) follows P.V(kt)).
    auto s_tile = [&](int kt, f32x16& s) {
      // S^T tile: 32 keys x 32 queries.  K fragments: one ds_read_b128 per four MFMAs, read by hand two groups ahead (left to the
      // compiler each group was "read, wait, 4 MFMAs").  Chunk cg of row l32 lives at byte ka0 ^ (cg << 4), ka0 = row base |
      // ((8 half) ^ (l32 & 15)) << 4.  The additive key bias (0 / -inf) is the first product: A = bias of key l32 at k = 0,
      // B = 1 at k = 0 (0 + x is exact, -inf absorbs: the same values as adding it afterwards, without 16 LDS reads in the softmax).
      const unsigned ka0 = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Ks + (kt * 32 + l32) * 64) +
                           ((unsigned)((half * 8) ^ (l32 & 15)) << 4);
      f32x4 kf[2];
      float msv;
      asm volatile("ds_read_b32 %0, %1" : "=v"(msv) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Ms + kt * 32 + l32)) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(kf[0]) : "v"(ka0) : "memory");
      asm volatile("ds_read_b128 %0, %1" : "=v"(kf[1]) : "v"(ka0 ^ 16u) : "memory");
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(msv));
      f32x16 z;
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = 0.f;
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(half == 0 ? msv : 0.f, half == 0 ? 1.f : 0.f, z, 0, 0, 0);
#pragma unroll
      for (int cg = 0; cg < 8; ++cg) {
        if (cg + 1 < 8) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kf[cg & 1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[cg & 1]));
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg & 1].x, qraw[cg].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg & 1].y, qraw[cg].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg & 1].z, qraw[cg].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[cg & 1].w, qraw[cg].w, s, 0, 0, 0);
        if (cg + 2 < 8) asm volatile("ds_read_b128 %0, %1" : "=v"(kf[cg & 1]) : "v"(ka0 ^ ((unsigned)(cg + 2) << 4)) : "memory");
      }
    };
    // the softmax of one score tile in four pieces (so that they can sit between the MFMA groups of the previous tile's P.V):
    //   rows [r0, r1): causal mask + running tile max;  then max over the halves, new running max, alpha;
    //   rows [r0, r1): weights + their sum;              then (after the previous P.V): l, m, rescale of O
    struct Soft { float tmax, m_new, m_safe, alpha, psum; };
    auto soft_max_rows = [&](f32x16& s, Soft& sv, int r0, int r1) {
#pragma unroll
      for (int r = r0; r < r1; ++r) sv.tmax = fmaxf(sv.tmax, s[r]);
      asm volatile("" : "+v"(sv.tmax));   // (here, not sunk to the use)
    };
    auto soft_scale = [&](const RkState& st, Soft& sv) {
      sv.tmax = ff_halves_max(sv.tmax);
      sv.m_new = fmaxf(st.m, sv.tmax);
      sv.m_safe = (sv.m_new == -INFINITY) ? 0.f : sv.m_new;
      sv.alpha = ff_exp2(st.m - sv.m_safe);
      asm volatile("" : "+v"(sv.alpha), "+v"(sv.m_safe));
    };
    auto soft_exp_rows = [&](f32x16& s, Soft& sv, int r0, int r1) {
#pragma unroll
      for (int r = r0; r < r1; ++r) {
        const float pp = ff_exp2(s[r] - sv.m_safe);
        s[r] = pp;
        sv.psum += pp;
        asm volatile("" : "+v"(s[r]));    // (here: the optimiser sinks the exponentials to their use behind the P.V loop otherwise)
      }
    };
    auto soft_commit = [&](RkState& st, const Soft& sv) {
      st.l = st.l * sv.alpha + sv.psum;
      st.m = sv.m_new;
      if (!__all(sv.alpha == 1.0f)) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { st.o0[e] *= sv.alpha; st.o1[e] *= sv.alpha; }
      }
    };
    // O^T += V^T P^T for key tile kt (weights in p); with `soft`: the scores sn of tile kt + 1 become weights meanwhile.
    // The V values of two key rows (four reads) are in flight ahead of the four MFMAs that take them (left to the compiler every
    // MFMA had its own read-and-wait in front of it: the kernel sits at the register limit, loads sink to their uses).
    auto pv_tile = [&](int kt, const f32x16& p, const bool soft, f32x16& sn, RkState& st) {
      const unsigned va = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(Vs + (kt * 32 + 4 * half) * 64 + l32);
      float x[2][4];
      Soft sv;
      sv.tmax = -INFINITY; sv.psum = 0.f; sv.m_new = 0.f; sv.m_safe = 0.f; sv.alpha = 1.f;
      rk_read_v<0>(x[0], va);
#pragma unroll
      for (int pr = 0; pr < 8; ++pr) {   // pair pr: accumulator rows r = 2 pr, 2 pr + 1
        if (pr + 1 < 8) {
          rk_read_v_dyn(x[(pr + 1) & 1], va, pr + 1);
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[pr & 1][0]), "+v"(x[pr & 1][1]), "+v"(x[pr & 1][2]), "+v"(x[pr & 1][3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[pr & 1][0]), "+v"(x[pr & 1][1]), "+v"(x[pr & 1][2]), "+v"(x[pr & 1][3]));
        }
        st.o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[pr & 1][0], p[2 * pr], st.o0, 0, 0, 0);
        st.o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[pr & 1][1], p[2 * pr], st.o1, 0, 0, 0);
        st.o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[pr & 1][2], p[2 * pr + 1], st.o0, 0, 0, 0);
        st.o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[pr & 1][3], p[2 * pr + 1], st.o1, 0, 0, 0);
        if (soft) {   // (constant after inlining) the next tile's softmax, a slice per MFMA group
          if (pr < 2) soft_max_rows(sn, sv, 8 * pr, 8 * pr + 8);
          else if (pr == 2) soft_scale(st, sv);
          else if (pr < 7) soft_exp_rows(sn, sv, 4 * (pr - 3), 4 * (pr - 3) + 4);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (soft) soft_commit(st, sv);
    };
    auto process = [&](int qt, int kt0, int kt1, bool fetched, RkState& st) {
      if (!fetched) fetch_q(qt);
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) qraw[cc] *= qscale;   // in place: one register set for the queries
      st.m = -INFINITY; st.l = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st.o0[e] = 0.f; st.o1[e] = 0.f; }
      f32x16 sa, sb;
      s_tile(kt0, sa);
      {
        Soft sv;
        sv.tmax = -INFINITY; sv.psum = 0.f;
        soft_max_rows(sa, sv, 0, 16);
        soft_scale(st, sv);
        soft_exp_rows(sa, sv, 0, 16);
        soft_commit(st, sv);
      }
      for (int kt = kt0;;) {
        if (kt + 1 >= kt1) { pv_tile(kt, sa, false, sa, st); break; }
        s_tile(kt + 1, sb);
        pv_tile(kt, sa, true, sb, st);
        ++kt;
        if (kt + 1 >= kt1) { pv_tile(kt, sb, false, sb, st); break; }
        s_tile(kt + 1, sa);
        pv_tile(kt, sb, true, sa, st);
        ++kt;
      }
      if (vtail && kt1 == ntiles) {   // wave-uniform: this range ends with the query tile's last full key tile
        float sj[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // rows past the tail are rows of the same (loaded, finite) key tile; their score is replaced below.  ONE address
          // register + immediates: key & 15 = j < 4, so chunk (8 half + cg) ^ j = 8 half + (cg ^ j)
          const int key = ntiles * 32 + j;
          const float* kr = Ks + (ntiles * 32) * 64 + half * 32;
          float acc = 0.f;
#pragma unroll
          for (int cg = 0; cg < 8; ++cg) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + j * 64 + ((cg ^ j) << 2));
            acc += (kf.x * qraw[cg].x + kf.y * qraw[cg].y) + (kf.z * qraw[cg].z + kf.w * qraw[cg].w);
          }
          acc = ff_halves_sum(acc);                      // the two lane halves hold the two halves of the head dimension
          float v = acc + Ms[key];
          if (j >= tail) v = -INFINITY;
          sj[j] = v;
          tmax = fmaxf(tmax, v);
        }
        const float m_new = fmaxf(st.m, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = ff_exp2(st.m - m_safe);
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sj[j] = ff_exp2(sj[j] - m_safe); psum += sj[j]; }
        st.l = st.l * alpha + (half == 0 ? psum : 0.f);   // both halves hold the SAME tail weights: count them once
        st.m = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { st.o0[e] *= alpha; st.o1[e] *= alpha; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < tail) {   // block-uniform
            const float* vr = Vs + (ntiles * 32 + j) * 64 + 4 * half;   // O register e <-> d = (e&3) + 8*(e>>2) + 4*half
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const f32x4 a = *reinterpret_cast<const f32x4*>(vr + 8 * g4);
              const f32x4 b = *reinterpret_cast<const f32x4*>(vr + 32 + 8 * g4);
              st.o0[g4 * 4 + 0] += sj[j] * a.x; st.o0[g4 * 4 + 1] += sj[j] * a.y;
              st.o0[g4 * 4 + 2] += sj[j] * a.z; st.o0[g4 * 4 + 3] += sj[j] * a.w;
              st.o1[g4 * 4 + 0] += sj[j] * b.x; st.o1[g4 * 4 + 1] += sj[j] * b.y;
              st.o1[g4 * 4 + 2] += sj[j] * b.z; st.o1[g4 * 4 + 3] += sj[j] * b.w;
            }
          }
        }
      }
    };
    auto store_out = [&](int qt, const RkState& st) {
      bool qv;
      const size_t qrow = query_row(qt, qv);
      const float l_tot = ff_halves_sum(st.l);
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      if (qv) {
        float* op = d.o + qrow * d.ldo + h * FF_HEAD_DIM + 4 * half;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 a = {st.o0[g4 * 4 + 0] * inv, st.o0[g4 * 4 + 1] * inv, st.o0[g4 * 4 + 2] * inv, st.o0[g4 * 4 + 3] * inv};
          f32x4 b = {st.o1[g4 * 4 + 0] * inv, st.o1[g4 * 4 + 1] * inv, st.o1[g4 * 4 + 2] * inv, st.o1[g4 * 4 + 3] * inv};
          ff_st16(op + 8 * g4, a);
          ff_st16(op + 32 + 8 * g4, b);
        }
      }
    };

    // A wave's range may begin inside a query tile (FIRST partial) and end inside another (LAST partial).  The last one stays in
    // registers until the barrier; the first one leaves for the wave's record in global memory at once (fire-and-forget 16-byte
    // stores, read back by the merging wave of the same block after the barrier): holding it cost 34 registers for the whole item
    // loop, which the second score register set of the software pipeline needs.
    RkState cur;
    float* const grec = scratch + (size_t)(blockIdx.x * RK_NW + wave) * RK_REC;
    bool has_last = false;
    int last_k = -1, last_kt0 = 0;
    for (int it = ia; it < ib;) {   // wave-uniform control flow
      const int k = it / ntiles, kt0 = it - k * ntiles;
      const int kt1 = (ntiles - kt0) < (ib - it) ? ntiles : kt0 + (ib - it);
      process(rank + k * c, kt0, kt1, it == ia, cur);
      it += kt1 - kt0;
      if (kt0 == 0 && kt1 == ntiles) {
        store_out(rank + k * c, cur);
      } else if (it < ib) {          // more items follow: this is the wave's FIRST partial
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          *reinterpret_cast<f32x4*>(grec + (g4 * 64 + lane) * 4) = f32x4{cur.o0[g4 * 4], cur.o0[g4 * 4 + 1], cur.o0[g4 * 4 + 2], cur.o0[g4 * 4 + 3]};
          *reinterpret_cast<f32x4*>(grec + ((4 + g4) * 64 + lane) * 4) = f32x4{cur.o1[g4 * 4], cur.o1[g4 * 4 + 1], cur.o1[g4 * 4 + 2], cur.o1[g4 * 4 + 3]};
        }
        grec[32 * 64 + lane] = cur.m;
        grec[33 * 64 + lane] = cur.l;
      } else {                       // the wave's LAST partial stays in `cur`
        has_last = true; last_k = k; last_kt0 = kt0;
      }
    }
    FF_EXP_ASTAMP(2);
#ifdef FF_EXP_ATTN_STAMP
    if (blockIdx.x == 0 && lane == 0) { ff_exp_attn_stamps[8 + wave] = __builtin_readcyclecounter(); ff_exp_attn_stamps[16 + wave] = (unsigned long long)(ib - ia); }
#endif
    __syncthreads();                 // every wave is done with K / V: the area now carries the partial records
    FF_EXP_ASTAMP(3);
    if (has_last) {                  // the last partial: LDS record of this wave
      float* rec = lds + wave * RK_REC;
#pragma unroll
      for (int e = 0; e < 16; ++e) { rec[e * 64 + lane] = cur.o0[e]; rec[(16 + e) * 64 + lane] = cur.o1[e]; }
      rec[32 * 64 + lane] = cur.m;
      rec[33 * 64 + lane] = cur.l;
    }
    __syncthreads();                 // (also orders the first partials' global stores before the reads below: same workgroup)
    if (has_last && last_kt0 == 0) {  // this wave holds the first keys of a cut query tile: merge in ascending key order
      const int kend = (last_k + 1) * ntiles;
      // the piece of wave w2 inside this tile is its FIRST partial (global record) if its range goes on past the tile
      auto grec_of = [&](int w2) { return scratch + (size_t)(blockIdx.x * RK_NW + w2) * RK_REC; };
      float m_star = cur.m;
      for (int w2 = wave + 1; w2 < RK_NW; ++w2) {
        int a2, b2;
        range_of(w2, a2, b2);
        if (a2 >= kend) break;
        if (b2 == a2) continue;
        const float mj = b2 > kend ? grec_of(w2)[32 * 64 + lane] : lds[w2 * RK_REC + 32 * 64 + lane];
        m_star = fmaxf(m_star, mj);
      }
      const float ms = (m_star == -INFINITY) ? 0.f : m_star;
      const float sc0 = ff_exp2(cur.m - ms);
      cur.l *= sc0;
#pragma unroll
      for (int e = 0; e < 16; ++e) { cur.o0[e] *= sc0; cur.o1[e] *= sc0; }
      for (int w2 = wave + 1; w2 < RK_NW; ++w2) {
        int a2, b2;
        range_of(w2, a2, b2);
        if (a2 >= kend) break;
        if (b2 == a2) continue;
        if (b2 > kend) {
          const float* rec = grec_of(w2);
          const float scj = ff_exp2(rec[32 * 64 + lane] - ms);
          cur.l += rec[33 * 64 + lane] * scj;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(rec + (g4 * 64 + lane) * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(rec + ((4 + g4) * 64 + lane) * 4);
            cur.o0[g4 * 4 + 0] += a.x * scj; cur.o0[g4 * 4 + 1] += a.y * scj; cur.o0[g4 * 4 + 2] += a.z * scj; cur.o0[g4 * 4 + 3] += a.w * scj;
            cur.o1[g4 * 4 + 0] += b.x * scj; cur.o1[g4 * 4 + 1] += b.y * scj; cur.o1[g4 * 4 + 2] += b.z * scj; cur.o1[g4 * 4 + 3] += b.w * scj;
          }
        } else {
          const float* rec = lds + w2 * RK_REC;
          const float scj = ff_exp2(rec[32 * 64 + lane] - ms);
          cur.l += rec[33 * 64 + lane] * scj;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            cur.o0[e] += rec[e * 64 + lane] * scj;
            cur.o1[e] += rec[(16 + e) * 64 + lane] * scj;
          }
        }
      }
      cur.m = m_star;
      store_out(rank + last_k * c, cur);
    }
    FF_EXP_ASTAMP(4);
    __syncthreads();                 // the next pair overwrites the area
    FF_EXP_ASTAMP(5);
  }
}

int g_attention_algo = 0;  // 0 = automatic, 1 = block-shared LDS staging, 2 = wave-independent

}  // namespace

extern "C" int ff_set_attention_algo(int algo) {
  const int old = g_attention_algo;
  g_attention_algo = algo;
  ff_tuning_changed();
  return old;
}

extern "C" int ff_attention(const ff_attn_desc* desc, ff_stream_t stream) {
  FF_CHECK_ARG(desc != nullptr, "ff_attention: null descriptor");
  const ff_attn_desc d = *desc;
  if (d.num_groups == 0 || d.nq == 0) return FF_OK;
  FF_CHECK_ARG(d.num_groups > 0 && d.num_heads > 0 && d.nq > 0 && d.nk >= 0, "ff_attention: bad counts");
  FF_CHECK_ARG(d.q && d.k && d.v && d.o, "ff_attention: null tensor");
  FF_CHECK_ARG(((d.ldq | d.ldk | d.ldv | d.ldo) & 3) == 0 && ff_aligned16(d.q) && ff_aligned16(d.k) &&
                   ff_aligned16(d.v) && ff_aligned16(d.o),
               "ff_attention: tensors must be 16-byte aligned with ld %% 4 == 0");
  const int width = d.num_heads * FF_HEAD_DIM;
  FF_CHECK_ARG(d.ldq >= width && d.ldk >= width && d.ldv >= width && d.ldo >= width,
               "ff_attention: ld smaller than num_heads*64");
  FF_CHECK_ARG(d.q_inner > 0, "ff_attention: q_inner must be positive");
  hipStream_t st = (hipStream_t)stream;
  const long gh = (long)d.num_groups * d.num_heads;
  int nw = d.nq > 64 ? 4 : (d.nq > 32 ? 2 : 1);
  const int q_tiles = ff_cdiv(d.nq, 32 * nw);
  const long blocks = gh * q_tiles;
  FF_CHECK_ARG(blocks < 2147483647L, "ff_attention: grid too large");
  FFProfScope prof(FF_CAT_ATTN, 4.0 * FF_HEAD_DIM * (double)gh * d.nq * d.nk, st);
  // automatic: the wave-independent kernel (with key splitting) wins while the launch cannot fill the
  // chip (measured crossover on MI355X: ~1500 units of 32 queries) and for the per-sequence self-attention
  // (at most two query tiles per group: the block-shared kernel then runs one- or two-wave blocks whose LDS
  // chunk limits a CU to four of them); above that the block-shared LDS staging moves 4x fewer bytes from L2
  // and is faster.
  // K/V-resident kernel: key sets of at most 288 rows shared by several query tiles (decoder cross-attention of the
  // 64..256-edge configurations, encoder); not for the per-sequence self-attention (2048+ tiny key sets)
  const int qt32 = ff_cdiv(d.nq, 32);
  // (the pair's K/V load is a fixed cost of every block: it pays from ~2 query tiles per CU on -- measured: config B
  //  from t = 8, never for the single-sequence decode, whose 8 pairs have at most 9 query tiles each)
  const bool resident_ok = d.nk <= RK_KEYS && d.nk > 0 && !d.causal;   // (cross-attention and encoder: no causal mask in that kernel)
  // 2 x fp16 kernel (round 6): the caller split K | V into fp16 planes (ff_attention_split_kv) -- same launches as the K/V-resident
  // kernel, an item at 24 half-length MFMAs instead of 65
  if (d.kv_planes && ff_attention_x2h_ok(d) &&
      (g_attention_algo == 4 || (g_attention_algo == 0 && ff_knob(FF_K_X2H_ATTN) && qt32 >= 4 && gh * qt32 >= 512)))
    return ff_attention_x2h_launch(d, d.kv_planes, (long long)ff_attention_planes_stride(), st);
  if (g_attention_algo == 3 ? resident_ok : (g_attention_algo == 0 && resident_ok && qt32 >= 4 && gh * qt32 >= 512)) {
    static std::atomic<bool> attr_done[16] = {};   // idempotent per-device attribute; host threads may race here
    int dev = 0;
    FF_CHECK_HIP(hipGetDevice(&dev));
    constexpr int lds_bytes = RK_LDS_FLOATS * (int)sizeof(float);
    if (dev < 0 || dev >= 16 || !attr_done[dev].load(std::memory_order_acquire)) {
      FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_resident_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      if (dev >= 0 && dev < 16) attr_done[dev].store(true, std::memory_order_release);
    }
    const int P = (int)gh;
    const int cus = ff_num_cus();
    int c = P <= cus ? cus / P : 1;
    if (c > qt32) c = qt32;
    if (c < 1) c = 1;
    const int nblocks = P <= cus ? P * c : cus;
    const int w_old = ff_knob(FF_K_RK_SPLIT_OLD);      // (probe knobs: tools/run_r04_attn.sh)
    const int w_young = ff_knob(FF_K_RK_SPLIT_YOUNG);
    const int phase_knob = ff_knob(FF_K_RK_PHASE);
    const int rot_on = ff_knob(FF_K_RK_ROTATE);   // (probe: 0 = every block loads K / V in the same order)
    float* scratch = nullptr;   // first partial records: one per wave
    FF_RETURN_IF(ff_stream_scratch(st, (size_t)nblocks * RK_NW * RK_REC * sizeof(float), &scratch));
    hipLaunchKernelGGL(attention_resident_kernel, dim3(nblocks), dim3(64 * RK_NW), lds_bytes, st, d, P, c, qt32,
                       w_old > 0 ? w_old : 1, w_young > 0 ? w_young : 1, phase_knob, scratch, rot_on);
    FF_CHECK_LAUNCH();
    return FF_OK;
  }
  // (long key sets -- beyond the resident kernel's 288 rows: the 300 ... 1024-edge wireframes of the config-E mix -- cross over
  //  earlier: a wave walks all key tiles of its unit one after the other.  Measured, profiles/r04/attention_long_keys.txt: 512
  //  units of 516 / 1028 keys 39 / 63 us (wave) vs 51 / 89 (block-shared), 600 units of 304 keys 39 vs 31, 1024 units of 1028
  //  keys 114 vs 94, 1200 units of 304 keys 70 vs 54)
  const long wave_units_max = d.nk > RK_KEYS ? 576 : 1536;
  const bool use_wave = g_attention_algo == 2 ||
                        (g_attention_algo == 0 && (gh * ff_cdiv(d.nq, 32) < wave_units_max || d.nq <= 64));
  if (use_wave) {
    // wave-independent kernel: units = (group, head, 32-query tile); the key tiles are dealt round-robin to ks waves
    // (ks = 1, 2, 4, 8: a power of two up to the tile count, idle waves allowed) while the launch would otherwise
    // leave SIMDs idle -- a wave's tiles are a serial chain of memory round trips.
    // short tails (see the kernel): keys / queries 1..8 past a multiple of 32 ride along with the full tile when every
    // unit is one wave anyway (ks == 1) and the key set is one tile plus the tail
    const bool tails = d.nk <= 40 && gh * (d.nq / 32) * 2 > 2048;
    const int qrem = d.nq & 31;
    const int qtail = (tails && d.nq > 32 && qrem >= 1 && qrem <= 8) ? qrem : 0;
    const int qt = qtail ? d.nq / 32 : ff_cdiv(d.nq, 32);
    const long units = gh * qt;
    const int key_tiles = ff_cdiv(d.nk, 32);
    int ks = 1;
    while (ks < 8 && units * ks * 2 <= 2048 && ks < key_tiles) ks *= 2;
    const int tail_ok = (tails && ks == 1) ? 1 : 0;
    if (ks == 8) {
      FF_CHECK_ARG(units < 2147483647L, "ff_attention: grid too large");
      hipLaunchKernelGGL((attention_wave_kernel<8>), dim3((unsigned)units), dim3(512), 0, st, d, qt, ks, units, 0, 0);
    } else {
      const long nblocks = (units + (4 / ks) - 1) / (4 / ks);
      FF_CHECK_ARG(nblocks < 2147483647L, "ff_attention: grid too large");
      hipLaunchKernelGGL((attention_wave_kernel<4>), dim3((unsigned)nblocks), dim3(256), 0, st, d, qt, ks, units, tail_ok,
                         qtail);
    }
    FF_CHECK_LAUNCH();
    return FF_OK;
  }
  if (nw == 4)
    hipLaunchKernelGGL((attention_kernel<4, true>), dim3((unsigned)blocks), dim3(256), 0, st, d, q_tiles);
  else if (nw == 2)
    hipLaunchKernelGGL((attention_kernel<2, false>), dim3((unsigned)blocks), dim3(128), 0, st, d, q_tiles);
  else
    hipLaunchKernelGGL((attention_kernel<1, false>), dim3((unsigned)blocks), dim3(64), 0, st, d, q_tiles);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
