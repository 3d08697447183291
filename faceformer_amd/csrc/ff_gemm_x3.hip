// fp32-accurate dense projection on the bf16 matrix cores ("3 x bf16"):
//     C = act(LN?(A) * W^T + bias [+ table]) + residual,   A fp32 [M,K], W fp32 [N,K] given as three bf16 planes.
//
// Every fp32 value is split EXACTLY into three bf16 terms, x = x1 + x2 + x3 (round-to-nearest each
// time, |x - x1 - x2 - x3| <= 2^-25 |x|), and a product is evaluated as the six partial products whose
// weight is >= 2^-16:  x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1); what is dropped is <= 2^-24 |xy|,
// the size of one fp32 rounding.  bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result carries the error of an ordinary fp32 dot product (measured against
// fp64: tests/test_hip_ops.py::test_gemm_x3_*) while the matrix pipe runs at 2.5 PF/s / 6 = 417 TF/s
// fp32-equivalent instead of the 157 TF/s of v_mfma_f32_32x32x2_f32.
//
// Plane layout of W: [plane][k / 16][row][k % 16]: a 16-wide K slice of 32 rows is one contiguous 1 KB block.
// Weights are split once (ff_split_weight_bf16x3, at model bind time).
//
// Kernel (round 4; tools/ubench/x3v2.hip is the probe it grew from):
//   * BOTH operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write, three slices in
//     flight): the weight planes as they are, the activations as fp32 rows.  A wave reads ITS OWN 32 rows back and splits
//     them into the three bf16 terms in registers (44 VALU per 16-wide slice, dealt out over the MFMA gaps); the two
//     waves that share 32 rows both split them: VALU work is not what limits this loop (see below).  Bank swizzles are
//     applied to the SOURCE address of the DMA (the LDS image of a piece is lane-linear).
//   * Transposed accumulators: D[n][m] = W fragment x A fragment, i.e. a lane owns ONE output row and groups of four
//     consecutive columns: the epilogue moves 16-byte pieces (16 stores per lane and 128 x 128 tile instead of 64), a
//     row's LayerNorm statistics are two lanes apart, and the LayerNorm of the consumer is two registers per lane.
//   * Fragments of slice s + 1 are read (one ds_read_b128 per MFMA gap) while the MFMAs of slice s issue; the second
//     fragment set fits because nothing is staged through registers.
//   * 64 x 128 block tiles: 2 x 2 waves of 32 x 64, 48 KB of LDS and at most 168 registers, i.e. three blocks per CU (two
//     with the LayerNorm-consuming form), and twice the tiles of a 128 x 128 tiling for the N = 512 projections of a
//     256-sequence decode step.  (The template also describes a 128-row tile, 4 x 1 waves of 32 x 128; it is not
//     instantiated: see x3_launch.)
//   * LayerNorm folding like the f32 family (ff_gemm_x3_ln): MODE 1 normalises the rows it splits ((x - mean) * rstd with
//     the row statistics merged from the producer's segment statistics, which arrive in an LDS patch by DMA one tile
//     ahead), MODE 2 leaves (mean, M2) per row and 32-column segment of what it stores.
//   * Launch shapes as before: whole tiles, or equal ranges of 32-wide K units per block with the cut tiles' partial
//     accumulators handed over through sc1 accesses and summed by the owning block in ascending block order
//     (deterministic); chosen per problem by a cost model.
// Measured (profiles/r04): the loop runs into the chip's POWER limit, not its issue limit -- 245-272 TF/s fp32-equivalent
// on zero-filled operands, 170-200 on random ones (same binary); pre-split activation planes are worth 0-8 % and were
// dropped again.
#include <mutex>
#include <vector>

#include <atomic>

#include "ff_common.h"
#include "ff_device.h"

// Timing experiment (tools/x3_phase_probe.py, -DX3_EXP_STAMP): wave 0 of workgroup 0 of gemm_x3_kernel sums, over its K-loop iterations,
// the shader-clock time of (0) the MFMA block with everything placed in its gaps, (1) the wait for the DMA / fragment reads behind
// it, (2) the segment-end branch + block barrier, (3) the rest of the iteration; [4] = iterations, [5] = whole kernel.
#ifdef X3_EXP_STAMP
__device__ unsigned long long ff_exp_x3_stamps[8];
extern "C" int ff_exp_read_x3_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_exp_x3_stamps), sizeof(ff_exp_x3_stamps)) == hipSuccess ? 0 : -1;
}
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct X3Args {
  const float* A;
  const float* A2;
  const unsigned short* Wp;  // [3][K/16][N][16] bf16 (see "plane layout" above)
  const float* W;            // gemm_dma_f32_kernel: the fp32 weight rows themselves
  int ldw;
  const float* bias;
  const float* res;
  float* C;
  int lda, ldr, ldc;
  int M, N, K;
  int n_split, act;
  int nt;                    // gemm_x3_kernel: terms per operand -- 3 (bf16 planes) or 2 (fp16 planes, ff_gemm_x2h)
  int tiles_m, tiles_n;
  long long plane_stride;  // elements between two planes of Wp
  int w_rows, w_row0;      // the planes describe a [w_rows, K] weight; the product uses rows [w_row0, w_row0 + N)
  int bal;                 // whole tiles: the blocks that get one tile more are dealt out over the XCDs and CUs
  // hybrid launch (hyb = 1): hw whole tiles per CU on `ha` of its block slots + the remaining tiles cut into hs K-pieces
  // each, one piece per CU on a further slot; cus = CUs the plan was made for, nA = blocks that run whole tiles
  int hyb, hw, hs, ha, cus, nA;
  float* ws;               // [grid][BM * 128] partial accumulators in register order
  unsigned int* flags;     // [grid]: 1 = slot holds a partial tile
  int upt;                 // units (32 k) per tile
  int base, rem, gran;     // block lb owns (base + (lb < rem)) allotments of `gran` units
  // LayerNorm folding (MODE 1 / 2), as GemmArgs of the f32 family
  const float* ln_in;      // MODE 1: [M][16][2] (mean, M2 over 32 columns) of the A rows (K = 512)
  float ln_eps;
  const float* rowtab;     // MODE 1: C[m][n] += rowtab[(m / rowtab_div) * ld_rowtab + n] for n < rowtab_cols
  int ld_rowtab, rowtab_div, rowtab_cols;
  float* ln_out;           // MODE 2: [M][N/32][2] segment statistics of the stored C rows
  const float* colsum;     // MODE 3: [w_rows] row sums of the folded weight (sum_k W'[n][k])
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

// fp16 terms of the same layout ([plane][k / 16][row][k % 16], two planes): w1 = fp16(w), w2' = fp16((w - w1) 2^11)
__global__ void split_weight_fp16_kernel(const float* __restrict__ W, int ldw, int N, int K, unsigned short* __restrict__ P) {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const size_t n2 = (size_t)N * (K / 2);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / (K / 2), c = (i % (K / 2)) * 2;
    const float x0 = W[row * ldw + c], x1 = W[row * ldw + c + 1];
    const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
    const f16x2 l = __builtin_convertvector(f32x2{(x0 - (float)h[0]) * 2048.0f, (x1 - (float)h[1]) * 2048.0f}, f16x2);
    unsigned* o = reinterpret_cast<unsigned*>(P);
    const size_t plane = (size_t)N * K / 2;
    const size_t at = ((c >> 4) * (size_t)N + row) * 8 + ((c & 15) >> 1);
    o[at] = __builtin_bit_cast(unsigned, h);
    o[plane + at] = __builtin_bit_cast(unsigned, l);
  }
}

#define X3_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// A ds_read_b128 the compiler does not see: it would put s_waitcnt vmcnt(0) in front of every LDS access that may alias
// the destination of an LDS-DMA in flight.  Results are fenced by the explicit lgkmcnt waits of the loop.
template <int OFF = 0>
__device__ __forceinline__ u32x4 x3_lds_read16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// W fragment R = plane * NI + ni of the slot whose (wave's) W fragments start at `addr`
template <int NI, int R>
__device__ __forceinline__ void x3_read_w(u32x4 (&wf)[3][NI], unsigned addr) {
  wf[R / NI][R % NI] = x3_lds_read16<(R / NI) * 4096 + (R % NI) * 1024>(addr);
}
template <int NI>
__device__ __forceinline__ void x3_read_w_i(u32x4 (&wf)[3][NI], unsigned addr, int r) {   // r: a constant after unrolling
  switch (r) {
    case 0: x3_read_w<NI, 0>(wf, addr); break; case 1: x3_read_w<NI, 1>(wf, addr); break;
    case 2: x3_read_w<NI, 2>(wf, addr); break; case 3: x3_read_w<NI, 3>(wf, addr); break;
    case 4: x3_read_w<NI, 4>(wf, addr); break; case 5: x3_read_w<NI, 5>(wf, addr); break;
    case 6: if (NI == 4) x3_read_w<NI, (NI == 4 ? 6 : 0)>(wf, addr); break;
    case 7: if (NI == 4) x3_read_w<NI, (NI == 4 ? 7 : 0)>(wf, addr); break;
    case 8: if (NI == 4) x3_read_w<NI, (NI == 4 ? 8 : 0)>(wf, addr); break;
    case 9: if (NI == 4) x3_read_w<NI, (NI == 4 ? 9 : 0)>(wf, addr); break;
    case 10: if (NI == 4) x3_read_w<NI, (NI == 4 ? 10 : 0)>(wf, addr); break;
    default: if (NI == 4) x3_read_w<NI, (NI == 4 ? 11 : 0)>(wf, addr); break;
  }
}
// plain (unpacked) VALU: the SLP pass would pair these into v_pk_add_f32 / v_pk_mul_f32, which issue slower beside MFMAs
__device__ __forceinline__ float x3_fsub(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float x3_fmul(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

constexpr int X3_BN = 128, X3_BK = 16;
// Depth of the LDS ring of K slices.  Slice s + RING is requested while slice s computes and must have landed one iteration
// before its fragments are read, so RING - 1.5 iterations of a block cover the DMA round trip.  Round 6 asked whether the fp16 loop
// (3 x 2 x 32 = 192 matrix-core cycles per slice and wave against 6 x 64 / 16 x 64 of the bf16 / f32 loops; MFMA-busy 0.18-0.28
// whatever the size) waits for its DMA: rings of four (plain forms, three blocks per CU) / five / six slots (the two-block
// LayerNorm consumers) measured the same as three within the run-to-run noise (profiles/r06/x2h_loop_probes.txt), and so did
// a loop that never waits for vmcnt -- the DMA round trip is covered.  Three slots stay; the depth is a build-time constant
// for probes (tools/build_variant.sh ... -DX3_RING_H=4).
#ifndef X3_RING_H
#define X3_RING_H 3
#endif
#ifndef X3_RING_H1
#define X3_RING_H1 3
#endif
#ifndef X3_RING_H3
#define X3_RING_H3 3
#endif
#ifndef X3_RING_F32
#define X3_RING_F32 3
#endif
template <int MODE, int NT>
constexpr int x3_ring() { return NT == 3 ? 3 : (MODE == 1 ? X3_RING_H1 : (MODE == 3 ? X3_RING_H3 : X3_RING_H)); }
constexpr int X3_STAT_BYTES = 16384;   // MODE 1: one 4 KB patch per wave (32 rows x 16 segments x (mean, M2))

// BM: 128 (4 x 1 waves) or 64 (2 x 2 waves).  MODE: 0 plain, 1 LayerNorm consumer (rows normalised before the split),
// 2 statistics producer, 3 LayerNorm consumer with the normalisation applied in the EPILOGUE (see x3_ln_linear).
// NT: terms per operand.  3 = the bf16 split above (six products).  2 = the fp16 split of round 6 ("2 x fp16", ff_gemm_x2h):
// x = x1 + x2 with x1 = fp16(x), x2' = fp16((x - x1) 2^11) (the second term is stored scaled by 2^11: it then has the magnitude of
// the first and stays out of fp16's subnormal range), 22 mantissa bits, THREE products x1 y1 + (x1 y2' + x2' y1) 2^-11 on
// v_mfma_f32_32x32x16_f16 -- half the matrix-core work (and energy: the bf16 form is power-limited) for an error that stays in
// the fp32 class (profiles/r06/fp16_split_error_table.txt; tests/test_hip_ops.py::test_gemm_x2h_*).  fp16 has 5 exponent
// bits: |x| must stay below 65504, which the callers guarantee (LayerNorm output is bounded by sqrt(K); the engine checks the
// norm bounds of the other operands when it binds the planes: faceformer_amd/hip/engine.py).
#ifndef X3_LN_H_BLOCKS      // blocks per CU of the LayerNorm-consuming fp16 form (MODE 1, NT 2).  Round 6 tried 3 (168 registers, spills on the
#define X3_LN_H_BLOCKS 2    // tile-change paths only; 52 KB of LDS each): +5 % on the isolated launch, -1.5 % on config B / C128 end to end -- stays at 2 (199 registers)
#endif
template <int MODE, int NT>
constexpr int x3_blocks_per_cu() { return MODE == 3 ? 2 : (MODE == 1 ? (NT == 2 ? X3_LN_H_BLOCKS : 2) : 3); }
template <int BM, int MODE, int NT>
__global__ __launch_bounds__(256, (x3_blocks_per_cu<MODE, NT>())) void gemm_x3_kernel(X3Args g) {   // (MODE 3 with two terms at three blocks per CU: 368 B of scratch, 10 accesses inside the MFMA runs -- stays at two)
  constexpr int BN = X3_BN, BK = X3_BK;
  constexpr int WN = 128 / BM;              // waves along N: 1 or 2
  constexpr int NI = BN / WN / 32;          // 32-column accumulators per wave: 4 or 2
  constexpr int NPA = BM / 64;              // A pieces (16 rows x 64 B) per wave and slice
  constexpr int NP = NPA + NT;              // DMA pieces per wave and slice
  constexpr int A_REG = BM * 64, SLOT = A_REG + NT * BN * 32;
  constexpr int RING = x3_ring<MODE, NT>();
  constexpr int NPROD = NT == 3 ? 6 : 3;    // partial products per fp32 product
  constexpr int NMF = NPROD * NI;           // MFMAs per wave and slice
  constexpr int NRD = 2 + NT * NI;          // fragment reads per wave and slice
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // ---- this block's unit range and its segments of the flat (tile, slice) sequence ----
  // contributed beginning part of the LAST tile first (kind 1: hand the raw accumulators over), then whole tiles (kind 0),
  // then the owned end part of the FIRST tile (kind 2: add the partials of the lower-numbered blocks, finish the tile)
  const int G = gridDim.x;
  const int upt = g.upt;
  int lb, u0, u1;
  if (g.hyb) {
    // Hybrid: block ids are dealt to the CUs round-robin (id % CUs): the first nA ids run the whole tiles of "their" CU
    // (tiles [cu * hw, (cu + 1) * hw), shared between the CU's `ha` whole-tile blocks), the others one K-piece of a remaining
    // tile each.  A CU then carries hw + 1 / hs tiles instead of hw + 1, and the pieces' exchange overlaps its whole tiles.
    lb = blockIdx.x;
    if (lb < g.nA) {
      const int cu = lb % g.cus, slot = lb / g.cus;
      const int first = cu * g.hw + (slot ? (g.hw + 1) / 2 : 0);
      const int cnt = g.ha == 1 ? g.hw : (slot ? g.hw / 2 : (g.hw + 1) / 2);
      u0 = first * upt; u1 = u0 + cnt * upt;
    } else {
      const int j = lb - g.nA, ups = upt / g.hs;
      u0 = (g.cus * g.hw + j / g.hs) * upt + (j % g.hs) * ups; u1 = u0 + ups;
    }
  } else {
    lb = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // which blocks get one allotment more: the first `rem` logical blocks -- or (whole tiles, g.bal) the first rem / 8 blocks of
    // EVERY XCD, which the dispatcher puts on different CUs.  (Logical blocks are XCD-contiguous: with "the first rem" two of
    // the eight XCDs got all the extra tiles and a 1.25-round launch ran like a 2-round one.)
    int before = lb < g.rem ? lb : g.rem, mine = lb < g.rem ? 1 : 0;
    if (g.bal && (G & 7) == 0) {
      const int x = blockIdx.x & 7, i = blockIdx.x >> 3, r8 = g.rem >> 3, rx = g.rem & 7;
      const int lim = r8 + (x < rx ? 1 : 0);
      before = x * r8 + (x < rx ? x : rx) + (i < lim ? i : lim);
      mine = i < lim ? 1 : 0;
    }
    u0 = (lb * g.base + before) * g.gran;
    u1 = u0 + (g.base + mine) * g.gran;
  }
  if (u0 >= u1) return;
#if defined(X3_EXP_STAGGER)         // probe: blocks start in three phases, X3_EXP_STAGGER x 0.85 us apart (de-phases the tile ends -- and with
  for (int q = 0; q < (int)(blockIdx.x % 3) * X3_EXP_STAGGER; ++q) __builtin_amdgcn_s_sleep(32);   // them the chip-wide bursts of result stores)
#endif
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
  auto tile_origin = [&](int id, int& m0, int& n0) {
    const int r = id % tiles_mn;
    m0 = (r / g.tiles_n) * BM; n0 = (r % g.tiles_n) * BN;
  };

  // ---- loader: uniform base pointers (advance per slice) + per-lane BYTE offsets (change per tile) ----
  int ld_p = 0, ld_j = 0, ld_end = 0;
  unsigned a_off[2] = {0, 0}, w_off;   // (a fixed bound: an array of dependent size makes the DMA builtin's arguments type-dependent,
                                       //  and the host pass then drops the whole kernel instantiation without a diagnostic)
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  const size_t w_step = (size_t)g.w_rows * 32, w_pl = (size_t)g.plane_stride * 2;
  auto set_tile = [&](int id, int j0) {
    int m0, n0;
    tile_origin(id, m0, n0);
    const float* Asrc = (g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A;
    a_base = reinterpret_cast<const char*>(Asrc) + (size_t)j0 * (BK * 4);
    w_base = reinterpret_cast<const char*>(g.Wp) + (size_t)j0 * w_step;
#pragma unroll
    for (int q = 0; q < NPA; ++q) {   // piece = 16 rows x 64 B; lane: row lane / 4, 16-byte slot lane % 4 (swizzled by (row >> 2) & 3)
      const int lr = 16 * (wave * NPA + q) + (lane >> 2);
      int row = m0 + lr;
      row = row < g.M ? row : g.M - 1;
      a_off[q] = ((unsigned)row * g.lda + 4 * ((lane & 3) ^ ((lr >> 2) & 3))) * 4;
    }
    {                                 // piece = 32 rows x 32 B of one plane; lane: row lane / 2, slot lane % 2 (swizzled by (row >> 4) & 1)
      const int lr = 32 * wave + (lane >> 1);
      int row = n0 + lr;
      row = row < g.N ? row : g.N - 1;
      w_off = (unsigned)(g.w_row0 + row) * 32 + (((lane & 1) ^ ((lr >> 4) & 1)) * 16);
    }
  };
  auto issue_piece = [&](int k, int slot) {   // piece k of the slice the loader stands on -> ring slot `slot`
    unsigned char* base = lds + slot * SLOT;
#if defined(X3_EXP_NODMA)          // probe: the loop without its operand traffic (computes on whatever the LDS holds)
    if (g.M > 0) return;
#elif defined(X3_EXP_NODMA_A)      // probe: ... without the activation pieces only
    if (k < NPA) return;
#elif defined(X3_EXP_NODMA_W)      // probe: ... without the weight pieces only
    if (k >= NPA) return;
#endif
    if (k < NPA) {
      __builtin_amdgcn_global_load_lds(a_base + a_off[k < NPA ? k : 0], X3_LDS_PTR(base + (wave * NPA + k) * 1024), 16, 0, 0);
    } else {
      const int p = k - NPA;
      __builtin_amdgcn_global_load_lds(w_base + p * w_pl + w_off, X3_LDS_PTR(base + A_REG + p * (BN * 32) + wave * 1024), 16, 0, 0);
    }
  };
  auto advance = [&]() {
#if defined(X3_EXP_NOADVANCE)      // probe: the loader stays on its first slice (no pointer / segment bookkeeping per slice)
    if (g.M > 0) return;
#endif
    if (++ld_j == ld_end) {
      if (ld_p + 1 < nseg) {
        int tile, j0, n, kind;
        segment(++ld_p, tile, j0, n, kind);
        set_tile(tile, j0);
        ld_j = j0; ld_end = j0 + n;
      } else {
        ld_j = ld_end - 1;   // past the end: the last slice again (never read)
      }
    } else { a_base += BK * 4; w_base += w_step; }
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NP; ++k) issue_piece(k, slot);
    advance();
  };

  // ---- fragment addresses (bytes inside a slot) ----
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned fw = A_REG + (wn * NI) * 1024 + l32 * 32 + ((half ^ (l32 >> 4)) * 16);        // + plane * 4096 + ni * 1024
  const unsigned fa_r0 = (32 * wm + l32) * 64 + (((2 * half) ^ ((l32 >> 2) & 3)) * 16);         // floats k = 8 half .. + 3
  const unsigned fa_r1 = (32 * wm + l32) * 64 + (((2 * half + 1) ^ ((l32 >> 2) & 3)) * 16);

  u32x4 wf[2][3][NI];   // [set][plane][ni]
  u32x4 af[2][3];       // [set][plane]
  u32x4 ar[2];          // fp32 rows of the next slice (8 floats of this lane's row)
  // Two accumulator sets: the x1 y1 products go to `acc`, the five small partial products (weight <= 2^-8) to `accs`; the sets
  // meet once per tile.  The matrix core rounds after EVERY one of the 16 products of an instruction (measured: a K = 512
  // chain of all six products in one accumulator is 4x as far from fp64 as the f32-MFMA kernel's, i.e. 6 x 512 roundings at
  // the magnitude of the result against 512); this way `acc` sees exactly the f32 kernel's 512 and the roundings of `accs`
  // are 2^-8 as large.
  f32x16 acc[NI], accs[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[ni][e] = 0.f; accs[ni][e] = 0.f; }

  // ---- MODE 1: (mean, rstd) of this lane's row for the tile being computed and the next one ----
  float mean_c = 0.f, rstd_c = 1.f, mean_n = 0.f, rstd_n = 1.f;
  float mean_s = 0.f, rstd_s = 1.f;   // what the split in flight uses
  const unsigned stat_lds = lds0 + RING * SLOT + wave * 4096;
  auto stats_fetch = [&](int p) {     // DMA the segment statistics of segment p's rows (this wave's 32) into the wave's patch
    if (MODE != 1 || p >= nseg) return;
    int tile, j0, n, kind, m0, n0;
    segment(p, tile, j0, n, kind);
    tile_origin(tile, m0, n0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {     // piece = 8 rows x 128 B
      int row = m0 + 32 * wm + 8 * q + (lane >> 3);
      row = row < g.M ? row : g.M - 1;
      // (16-byte chunk c of patch row r sits at position c ^ ((r >> 1) & 7): the merge's reads are then bank-conflict free)
      __builtin_amdgcn_global_load_lds(g.ln_in + (size_t)row * 32 + ((lane & 7) ^ ((4 * q + (lane >> 4)) & 7)) * 4,
                                       X3_LDS_PTR(lds + RING * SLOT + wave * 4096 + q * 1024), 16, 0, 0);
    }
  };
  auto stats_merge = [&](float& mean, float& rstd) {   // Chan's update for 16 equal parts (as ff_ln_finish of the f32 family)
    if (MODE != 1) return;
    // Row l32 of the patch, chunk c at position c ^ ((l32 >> 1) & 7) (see stats_fetch): the 16 lanes of a ds_read_b128 lane
    // group then touch 16 different 16-byte slots of the 256-byte bank row.  (Unswizzled, 128-byte row strides put 8 lanes of
    // a group on the same four banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40 for this kernel form in round 5.)
    u32x4 s[8];
    const unsigned srow = stat_lds + l32 * 128, skx = ((l32 >> 1) & 7) << 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = x3_lds_read16<0>(srow + ((16u * c) ^ skx));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7])::"memory");
    float sm = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, s[i]);
      sm += v[0] + v[2]; m2 += v[1] + v[3];
    }
    mean = sm * (1.0f / 16.0f);
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, s[i]);
      const float d0 = v[0] - mean, d1 = v[2] - mean;
      dev += d0 * d0 + d1 * d1;
    }
    const float var = (m2 + 32.f * dev) * (1.0f / 512.0f);
    rstd = 1.0f / sqrtf(var + g.ln_eps);
  };

  // ---- split of the 8 floats in ar[] in eight steps (pair q = step / 2, level = step % 2): dealt out over the MFMA gaps;
  //      the results are pinned where they are computed (the optimiser would sink them to their use) ----
  float r_[4][2];
  unsigned p1_[4], p2_[4], p3_[4];
  auto split_step = [&](int st) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int q = st >> 1;
    if ((st & 1) == 0) {
      const f32x4 v = __builtin_bit_cast(f32x4, ar[q >> 1]);
      float x0 = v[2 * (q & 1)], x1 = v[2 * (q & 1) + 1];
      if (MODE == 1) {   // (x - mean) * rstd, the reference's order of operations
        x0 = x3_fmul(x3_fsub(x0, mean_s), rstd_s);
        x1 = x3_fmul(x3_fsub(x1, mean_s), rstd_s);
      }
      if (NT == 2) {     // fp16 terms: x1 = fp16(x) (round to nearest), residual exact in fp32
        if (MODE == 3) {   // RAW (un-normalised) rows: 2^-6 keeps |x| up to 4.2e6 inside fp16's range (undone, exactly, per tile)
          x0 = x3_fmul(x0, 0.015625f);
          x1 = x3_fmul(x1, 0.015625f);
        }
#if defined(X3_EXP_NOSPLIT)      // probe: no split arithmetic at all (wrong numbers; the upper bound of what cheaper splitting can buy)
        p1_[q] = __builtin_bit_cast(unsigned, x0);
        r_[q][0] = x1; r_[q][1] = x0;
        asm volatile("" : "+v"(p1_[q]), "+v"(r_[q][0]), "+v"(r_[q][1]));
        return;
#elif defined(X3_MIXSPLIT)
        // x1 = fp16(x) (one packed conversion); the scaled inputs t = x 2^11 (exact); the second terms then are ONE mixed-precision
        // fma each, fp16(fma(x1, -2^11, t)) = fp16((x - x1) 2^11) -- fma's argument is exact in fp32, so the bits are those of the
        // five-instruction form (convert back, subtract, scale, convert) at five VALU per pair instead of eight
        const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
        p1_[q] = __builtin_bit_cast(unsigned, h);
        r_[q][0] = x3_fmul(x0, 2048.0f);
        r_[q][1] = x3_fmul(x1, 2048.0f);
        asm volatile("" : "+v"(p1_[q]), "+v"(r_[q][0]), "+v"(r_[q][1]));
        return;
#else
        const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
        p1_[q] = __builtin_bit_cast(unsigned, h);
        r_[q][0] = x3_fsub(x0, (float)h[0]);
        r_[q][1] = x3_fsub(x1, (float)h[1]);
        asm volatile("" : "+v"(p1_[q]), "+v"(r_[q][0]), "+v"(r_[q][1]));
        return;
#endif
      }
      p1_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
      r_[q][0] = x3_fsub(x0, __builtin_bit_cast(float, p1_[q] << 16));
      r_[q][1] = x3_fsub(x1, __builtin_bit_cast(float, p1_[q] & 0xffff0000u));
      asm volatile("" : "+v"(p1_[q]), "+v"(r_[q][0]), "+v"(r_[q][1]));
    } else if (NT == 2) {   // second term, scaled by 2^11 (exact)
#if defined(X3_EXP_NOSPLIT)
      p2_[q] = __builtin_bit_cast(unsigned, r_[q][0]);
      asm volatile("" : "+v"(p2_[q]));
#elif defined(X3_MIXSPLIT)
      unsigned d;
      const float m2048 = -2048.0f;
      asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
                   "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                   : "=&v"(d) : "v"(p1_[q]), "v"(m2048), "v"(r_[q][0]), "v"(r_[q][1]));
      p2_[q] = d;
      asm volatile("" : "+v"(p2_[q]));
#else
      p2_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{x3_fmul(r_[q][0], 2048.0f), x3_fmul(r_[q][1], 2048.0f)}, f16x2));
      asm volatile("" : "+v"(p2_[q]));
#endif
    } else {
      p2_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{r_[q][0], r_[q][1]}, bf16x2));
      const float s0 = x3_fsub(r_[q][0], __builtin_bit_cast(float, p2_[q] << 16));
      const float s1 = x3_fsub(r_[q][1], __builtin_bit_cast(float, p2_[q] & 0xffff0000u));
      p3_[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
      asm volatile("" : "+v"(p2_[q]), "+v"(p3_[q]));
    }
  };
  auto split_collect = [&](u32x4 (&dst)[3]) {
    dst[0] = u32x4{p1_[0], p1_[1], p1_[2], p1_[3]};
    dst[1] = u32x4{p2_[0], p2_[1], p2_[2], p2_[3]};
    if (NT == 3) dst[2] = u32x4{p3_[0], p3_[1], p3_[2], p3_[3]};
  };

  // ---- compute-side segment state ----
  int cp_p = 0, cp_cnt = 0, cp_n = 0, cp_kind = 0;
  int e_m0 = 0, e_n0 = 0;
  auto begin_segment = [&](int p) {
    int id, j0;
    segment(p, id, j0, cp_n, cp_kind);
    cp_cnt = 0;
    tile_origin(id, e_m0, e_n0);
  };

  // End of a segment: hand the partial tile over (kind 1), or finish the tile -- after adding the partials of the
  // lower-numbered blocks (kind 2) -- with bias / activation / table / residual, the store and (MODE 2) the statistics.
  // Every global access of this path is inline assembly with its own waits: accesses the compiler knows about leave an
  // "unknown" wait-count state at the join with the K loop (it then opens every slice with s_waitcnt vmcnt(0)), and the
  // stores must not be waited for at all -- they drain behind the next tile's first slices.
  auto gload16 = [&](const float* ptr) -> f32x4 {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
  };
  auto end_segment = [&]() {
    constexpr int NQ = 4 * NI;   // 16-byte accumulator groups per lane
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {   // (fp16 terms: the small products were accumulated at 2^11 times their weight)
        if (NT == 2 && MODE == 3) acc[ni][e] = acc[ni][e] * 64.0f + accs[ni][e] * (64.0f / 2048.0f);   // (... and the rows at 2^-6)
        else acc[ni][e] += NT == 2 ? accs[ni][e] * (1.0f / 2048.0f) : accs[ni][e];
        accs[ni][e] = 0.f;
      }
    if (cp_kind == 1) {
      f32x4* wp = reinterpret_cast<f32x4*>(g.ws + (size_t)lb * (BM * BN)) + tid;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + (ni * 4 + q) * 256), "v"(v) : "memory");
          acc[ni][4 * q] = 0.f; acc[ni][4 * q + 1] = 0.f; acc[ni][4 * q + 2] = 0.f; acc[ni][4 * q + 3] = 0.f;
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) __hip_atomic_store(g.flags + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (cp_kind == 2) {  // add the partials of the blocks that hold units [k0 * upt, u0) of this tile
      int c0;
      if (g.hyb) {
        c0 = lb - (g.hs - 1);               // the other pieces of this tile: the hs - 1 block ids below this one
      } else {
        const int ab = (k0 * upt) / g.gran;   // allotment that starts the tile (unit ranges: gran = 1)
        const int big = g.rem * (g.base + 1);
        c0 = ab < big ? ab / (g.base + 1) : g.rem + (ab - big) / g.base;
      }
      for (int c = c0; c < lb; ++c) {
        while (__hip_atomic_load(g.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const f32x4* rp = reinterpret_cast<const f32x4*>(g.ws + (size_t)c * (BM * BN)) + tid;
        constexpr int TQ = 8;   // 8 x 16 bytes in flight per lane and round trip (all 16 of a 128-row tile would spill)
#pragma unroll
        for (int h = 0; h < NQ / TQ; ++h) {
          f32x4 t[TQ];
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[j]) : "v"(rp + (h * TQ + j) * 256) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < TQ; ++j) {
            asm volatile("" : "+v"(t[j]));
            const int q = h * TQ + j;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[q >> 2][4 * (q & 3) + i] += t[j][i];
          }
        }
      }
      __builtin_amdgcn_s_barrier();  // every thread is past its flag polls
      if (tid < lb - c0) __hip_atomic_store(g.flags + c0 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // This lane: row m, columns n0 + (wn * NI + ni) * 32 + 8 q + 4 half + {0..3}.  ALL epilogue operands of the tile are
    // requested in one go (the fragments of the next slice are not live here: they are read again behind the epilogue),
    // every result is finished in its accumulator registers, then the stores go out back to back.  A lane reads and
    // writes the same elements, so a residual that aliases C stays correct.
    const int m = e_m0 + 32 * wm + l32;
    const bool rowok = m < g.M;
    const int mc = rowok ? m : g.M - 1;
    const int nb0 = e_n0 + wn * NI * 32 + 4 * half;
    const bool has_tab = (MODE == 1 || MODE == 3) && g.rowtab != nullptr;
    const float* xrow = nullptr;   // residual row, or (MODE 1) the row of the position table
    int xlim = 0;
    if (g.res) { xrow = g.res + (size_t)mc * g.ldr; xlim = g.N; }
    else if (has_tab) { xrow = g.rowtab + (size_t)(mc / g.rowtab_div) * g.ld_rowtab; xlim = g.rowtab_cols; }
    // MODE 3: LayerNorm(x) W'^T = rstd (x W'^T - mean colsum(W')): the K loop multiplied the RAW rows; this row's statistics are
    // merged here from the producer's 16 segment statistics (Chan's update, as stats_merge) and applied to the finished sums.
    float ln_mean = 0.f, ln_rstd = 1.f;
    f32x4 sv[MODE == 3 ? 8 : 1];
    if (MODE == 3) {   // (requested together with the first operand group below: one round trip instead of two)
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] = gload16(g.ln_in + (size_t)mc * 32 + 4 * i);
    }
    // a few 32-column groups at a time: with all of a 128-column row's operands in flight at once the register allocator
    // spills loop-invariant addresses INTO the K loop
    constexpr int GQ = (NI == 4 || MODE == 3) ? 4 : 8;
#pragma unroll
    for (int h = 0; h < NQ / GQ; ++h) {
      f32x4 xv[GQ], bv[GQ], cv[GQ];
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        const int q = h * GQ + j;
        const int n = nb0 + (q >> 2) * 32 + (q & 3) * 8;
        xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !defined(X3_EXP_NOEPILOAD)      // probe: the epilogue without its bias / residual / table loads (wrong numbers)
        if (xrow && n + 3 < xlim) xv[j] = gload16(xrow + n);
        if (g.bias && n + 3 < g.N) bv[j] = gload16(g.bias + n);
#endif
        if (MODE == 3 && n + 3 < g.N) cv[j] = gload16(g.colsum + g.w_row0 + n);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        asm volatile("" : "+v"(xv[j]), "+v"(bv[j]));
        if (MODE == 3) asm volatile("" : "+v"(cv[j]));
      }
      if (MODE == 3 && h == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(sv[i]));
        float sm = 0.f, m2 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sm += sv[i][0] + sv[i][2]; m2 += sv[i][1] + sv[i][3]; }
        ln_mean = sm * (1.0f / 16.0f);
        float dev = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d0 = sv[i][0] - ln_mean, d1 = sv[i][2] - ln_mean; dev += d0 * d0 + d1 * d1; }
        ln_rstd = 1.0f / sqrtf((m2 + 32.f * dev) * (1.0f / 512.0f) + g.ln_eps);
      }
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        const int q = h * GQ + j, ni = q >> 2, qq = q & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[ni][4 * qq + i];
          if (MODE == 3) v = (v - ln_mean * cv[j][i]) * ln_rstd;
          v += bv[j][i];
          if (MODE == 1 || MODE == 3) {   // position-table term in front of the activation, no residual
            v += xv[j][i];
            if (g.act == 1) v = fmaxf(v, 0.f);
          } else {
            if (g.act == 1) v = fmaxf(v, 0.f);
            v += xv[j][i];
          }
          acc[ni][4 * qq + i] = v;
        }
      }
    }
    if (MODE == 2) {   // (mean, M2) of this row's 32 stored values per column group: 16 here, 16 in the lane 32 away
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sm += acc[ni][e];
        sm = ff_halves_sum(sm);
        const float mean = sm * (1.0f / 32.0f);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = acc[ni][e] - mean; m2 += d * d; }
        m2 = ff_halves_sum(m2);
        const int seg = (e_n0 >> 5) + wn * NI + ni;
        if (half == 0 && rowok && seg * 32 < g.N) {
          const f32x2 st2 = f32x2{mean, m2};
          asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g.ln_out + ((size_t)m * (g.N >> 5) + seg) * 2), "v"(st2) : "memory");
        }
      }
    }
    if (rowok) {
      float* cp = g.C + (size_t)m * g.ldc + nb0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (nb0 + (q >> 2) * 32 + (q & 3) * 8 + 3 < g.N) {
          const f32x4 v = {acc[q >> 2][4 * (q & 3)], acc[q >> 2][4 * (q & 3) + 1], acc[q >> 2][4 * (q & 3) + 2], acc[q >> 2][4 * (q & 3) + 3]};
#if defined(X3_EXP_NOSTORE)        // probe: results are not stored (one store per lane and tile keeps the accumulators alive)
          if (q == 0)
#endif
#if defined(X3_EXP_NTSTORE)        // probe: non-temporal result stores
          asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(cp + (q >> 2) * 32 + (q & 3) * 8), "v"(v) : "memory");
#else
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(cp + (q >> 2) * 32 + (q & 3) * 8), "v"(v) : "memory");
#endif
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
  };

  // ---- prologue: slices 0, 1, 2 in flight; 0 and 1 landed; fragments of slice 0 in set 0 ----
  {
    int tile, j0, n, kind;
    segment(0, tile, j0, n, kind);
    set_tile(tile, j0);
    ld_j = j0; ld_end = j0 + n;
  }
  if (MODE == 1) {   // statistics of segment 0 (and 1): fetched and merged before the first split
    stats_fetch(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stats_merge(mean_c, rstd_c);
    stats_fetch(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stats_merge(mean_n, rstd_n);
    stats_fetch(2);
    mean_s = mean_c; rstd_s = rstd_c;
  }
#pragma unroll
  for (int r = 0; r < RING; ++r) issue(r);
  begin_segment(0);
  static_assert((RING - 2) * NP <= 15, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0F70 | ((RING - 2) * NP));   // vmcnt: slices 0 and 1 have landed, the younger ones may be in flight
  __builtin_amdgcn_s_barrier();
  {
    ar[0] = x3_lds_read16(lds0 + fa_r0);
    ar[1] = x3_lds_read16(lds0 + fa_r1);
#pragma unroll
    for (int r = 0; r < NT * NI; ++r) x3_read_w_i<NI>(wf[0], lds0 + fw, r);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0]), "+v"(ar[1])::"memory");
#pragma unroll
    for (int st = 0; st < 8; ++st) split_step(st);
    split_collect(af[0]);
  }
  __builtin_amdgcn_s_barrier();

  // (A plane, W plane) of the partial products, small terms first; the LAST one (x1 y1) has its own accumulator
  constexpr int PA[6] = {NT == 3 ? 2 : 1, NT == 3 ? 1 : 0, 0, 1, 0, 0}, PB[6] = {0, 1, NT == 3 ? 2 : 0, 0, 1, 0};
  // where the split steps and the DMA pieces go among the MFMA gaps
  constexpr int SP0 = NI == 4 ? 5 : 3;            // first split step (the two row reads are the oldest of SP0 + 1 reads)
  constexpr int SPG = (8 + (NMF - SP0) - 1) / (NMF - SP0);   // split steps per gap (1 with six products, 3 with three)
  constexpr int SPC = SP0 + (8 + SPG - 1) / SPG;  // gap behind which the split is complete
  constexpr int DM0 = NI == 4 ? NRD : NMF - NP;   // first DMA piece
  int s0 = 0, s1 = 1;   // ring slots of slice s, s + 1  (slice s + RING goes to slot s0)
  const int total = 2 * (u1 - u0);
#ifdef X3_EXP_STAMP
  const bool stamping = blockIdx.x == 0 && wave == 0;
  unsigned long long st_d[4] = {0, 0, 0, 0}, st_n = 0, st_t3 = 0;
  const unsigned long long st_begin = __builtin_readcyclecounter();
#endif
  for (int s = 0; s < total; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#ifdef X3_EXP_STAMP
      unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0;
      if (stamping) { st_t0 = __builtin_readcyclecounter(); if (st_n) st_d[3] += st_t0 - st_t3; }
#endif
      const unsigned nb = lds0 + s1 * SLOT;   // slot of slice s + 1
      if (MODE == 1) {   // the rows split in this iteration belong to the next segment's tile when this is the segment's last slice
        const bool nx = cp_cnt + 1 == cp_n;
        mean_s = nx ? mean_n : mean_c; rstd_s = nx ? rstd_n : rstd_c;
      }
      // NMF MFMAs of slice s; in their gaps: the reads of slice s + 1 (one per gap), the split of its rows, the DMA of s + 3
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
#if defined(X3_EXP_MFMA_ORDER)     // probe: the x1 y1 products between the two small ones (same-accumulator MFMAs four apart, not two)
        const int t = NT == 2 ? (i / NI == 1 ? 2 : (i / NI == 2 ? 1 : 0)) : i / NI, ni = i % NI;
#else
        const int t = i / NI, ni = i % NI;
#endif
        if (NT == 3) {
          if (t < 5) accs[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[u][PB[t]][ni]),
                                                                        __builtin_bit_cast(bf16x8, af[u][PA[t]]), accs[ni], 0, 0, 0);
          else acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[u][PB[t]][ni]),
                                                                 __builtin_bit_cast(bf16x8, af[u][PA[t]]), acc[ni], 0, 0, 0);
        } else {
#if defined(X3_EXP_NOMFMA)         // probe: the loop without its matrix-core work
          if (g.M < 0)
#endif
          if (t < 2) accs[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[u][PB[t]][ni]),
                                                                       __builtin_bit_cast(f16x8, af[u][PA[t]]), accs[ni], 0, 0, 0);
          else acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wf[u][PB[t]][ni]),
                                                                __builtin_bit_cast(f16x8, af[u][PA[t]]), acc[ni], 0, 0, 0);
        }
#if defined(X3_EXP_NOLDSREAD)       // probe: no fragment reads in the loop (both register sets keep the prologue's values)
        if (g.M < 0) {
#else
        {
#endif
        if (i == 0) ar[0] = x3_lds_read16(nb + fa_r0);
        else if (i == 1) ar[1] = x3_lds_read16(nb + fa_r1);
        else if (i < NRD) x3_read_w_i<NI>(wf[u ^ 1], nb + fw, i - 2);
        }
        if (i == SP0) {
          if (SP0 == 5) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ar[0]), "+v"(ar[1])::"memory");
          else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ar[0]), "+v"(ar[1])::"memory");
        }
        if (i >= SP0) {
#pragma unroll
          for (int k = 0; k < SPG; ++k)
            if ((i - SP0) * SPG + k < 8) split_step((i - SP0) * SPG + k);
        }
        if (i == SPC) split_collect(af[u ^ 1]);
        if (i >= DM0 && i < DM0 + NP) issue_piece(i - DM0, s0);
        if (i == NMF - 1) advance();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (SPC >= NMF) split_collect(af[u ^ 1]);
#ifdef X3_EXP_STAMP
      if (stamping) st_t1 = __builtin_readcyclecounter();
#endif
      // slice s + 2 has landed (own pieces), every fragment of slice s + 1 is in registers
#if defined(X3_EXP_NOVMWAIT)      // probe: the loop never waits for its DMA (reads whatever the LDS holds)
      __builtin_amdgcn_s_waitcnt(0x0F70 & ~0x0F00);   // lgkmcnt(0) only
#else
      __builtin_amdgcn_s_waitcnt(0x0070 | ((RING - 2) * NP));   // vmcnt: everything but the youngest RING - 2 slices; lgkmcnt(0)
#endif
#pragma unroll
      for (int p = 0; p < NT; ++p)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[u ^ 1][p][ni]));
#ifdef X3_EXP_STAMP
      if (stamping) st_t2 = __builtin_readcyclecounter();
#endif
      if (++cp_cnt == cp_n) {  // block-uniform: the last slice of the segment was just issued
        end_segment();
        if (++cp_p < nseg) begin_segment(cp_p);
        // 64-row tiles: the fragments of slice s + 1 are read AGAIN here (their slot is untouched until the barrier below),
        // so the values read inside the loop are dead across the epilogue and its operands do not compete with them for the
        // 168 registers of three waves per SIMD.  (128-row tiles keep them live: the second copy of the read-and-split code
        // costs that kernel more registers than it frees.)  Either way no register that an asynchronous ds_read has not
        // filled yet may be spilled: tools/check_x3_asm.py looks for scratch accesses inside the MFMA runs of every build.
        if (NI == 2) {
          ar[0] = x3_lds_read16(nb + fa_r0);
          ar[1] = x3_lds_read16(nb + fa_r1);
#pragma unroll
          for (int r = 0; r < NT * NI; ++r) x3_read_w_i<NI>(wf[u ^ 1], nb + fw, r);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0]), "+v"(ar[1])::"memory");
#pragma unroll
          for (int p = 0; p < NT; ++p)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[u ^ 1][p][ni]));
          mean_s = mean_n; rstd_s = rstd_n;   // (MODE 1: these rows belong to the segment that starts now)
#pragma unroll
          for (int st = 0; st < 8; ++st) split_step(st);
          split_collect(af[u ^ 1]);
        }
        if (MODE == 1) {       // statistics: the next segment's become current, the one after is merged from the patch
          mean_c = mean_n; rstd_c = rstd_n;
          stats_merge(mean_n, rstd_n);
          stats_fetch(cp_p + 2);
        }
      }
#if defined(X3_EXP_NOBARRIER)     // probe: no block barrier in the K loop (races: wrong numbers)
#elif defined(X3_EXP_HALFBARRIER) // probe: a barrier every second slice only
      if (u == 1) __builtin_amdgcn_s_barrier();
#else
      __builtin_amdgcn_s_barrier();
#endif
#if !defined(X3_EXP_FIXEDSLOT)      // probe: without the ring rotation (slot offsets become loop constants)
      s0 = s1; s1 = s1 + 1 == RING ? 0 : s1 + 1;
#endif
#ifdef X3_EXP_STAMP
      if (stamping) { st_t3 = __builtin_readcyclecounter(); st_d[0] += st_t1 - st_t0; st_d[1] += st_t2 - st_t1; st_d[2] += st_t3 - st_t2; ++st_n; }
#endif
    }
  }
#ifdef X3_EXP_STAMP
  if (stamping && lane == 0) {
    for (int q = 0; q < 4; ++q) ff_exp_x3_stamps[q] = st_d[q];
    ff_exp_x3_stamps[4] = st_n; ff_exp_x3_stamps[5] = __builtin_readcyclecounter() - st_begin;
  }
#endif
}

// ---- the same launch structure on the f32 matrix cores ------------------------------------------------------------------------
// gemm_dma_f32_kernel: v_mfma_f32_32x32x2_f32 fed the way gemm_x3_kernel is fed -- BOTH operands as fp32 rows by LDS-DMA (no
// staging registers, no ds_write), fragments of slice s + 1 read in the MFMA gaps of slice s, transposed accumulators, 16-byte
// epilogue accesses, whole tiles + hybrid remainder, the same LayerNorm-folded forms (MODE 1 normalises the eight floats of its
// row per slice in registers).  A 16-wide slice is 16 MFMAs of 64 cycles per wave against 6 fragment reads and 3 DMA pieces, 36 KB
// of LDS; three blocks per CU (at the 128 registers of four, the epilogue's operands push loop addresses into scratch).  W is the nn.Linear weight itself ([N, K] fp32 rows, ld = ldw): no planes.
// Operand k order inside a slice: MFMA e of 8 takes k = e from the lanes 0..31 and k = 8 + e from the lanes 32..63 of BOTH
// operands (a lane holds floats 8 half .. 8 half + 7 of its row).
// BN: 128 (2 x 2 waves of 32 x 64), or -- round 6 -- 64 (2 x 2 waves of 32 x 32: twice the tiles for the 512-column projections of
// the middle steps; 8 MFMAs, 4 fragment reads and 2 DMA pieces per wave and slice, 24 KB of LDS).
template <int BM, int MODE, int BN = X3_BN>
__global__ __launch_bounds__(256, 3) void gemm_dma_f32_kernel(X3Args g) {
  constexpr int BK = X3_BK;
  constexpr int WN = 128 / BM;              // waves along N: 1 or 2
  constexpr int NI = BN / WN / 32;          // 32-column accumulators per wave: 4 or 2
  constexpr int NPA = BM / 64;              // A pieces (16 rows x 64 B) per wave and slice
  constexpr int NPW = BN / 64;              // W pieces (16 rows x 64 B) per wave and slice
  constexpr int NP = NPA + NPW;             // DMA pieces per wave and slice
  constexpr int A_REG = BM * 64, SLOT = A_REG + BN * 64;
  constexpr int RING = X3_RING_F32;
  constexpr int NMF = 8 * NI;               // MFMAs per wave and slice
  constexpr int NRD = 2 + 2 * NI;           // fragment reads per wave and slice
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nsl = g.K / BK;
  const int tiles_mn = g.tiles_m * g.tiles_n;

  // ---- this block's unit range and its segments of the flat (tile, slice) sequence ----
  // contributed beginning part of the LAST tile first (kind 1: hand the raw accumulators over), then whole tiles (kind 0),
  // then the owned end part of the FIRST tile (kind 2: add the partials of the lower-numbered blocks, finish the tile)
  const int G = gridDim.x;
  const int upt = g.upt;
  int lb, u0, u1;
  if (g.hyb) {
    // Hybrid: block ids are dealt to the CUs round-robin (id % CUs): the first nA ids run the whole tiles of "their" CU
    // (tiles [cu * hw, (cu + 1) * hw), shared between the CU's `ha` whole-tile blocks), the others one K-piece of a remaining
    // tile each.  A CU then carries hw + 1 / hs tiles instead of hw + 1, and the pieces' exchange overlaps its whole tiles.
    lb = blockIdx.x;
    if (lb < g.nA) {
      const int cu = lb % g.cus, slot = lb / g.cus;
      const int first = cu * g.hw + (slot ? (g.hw + 1) / 2 : 0);
      const int cnt = g.ha == 1 ? g.hw : (slot ? g.hw / 2 : (g.hw + 1) / 2);
      u0 = first * upt; u1 = u0 + cnt * upt;
    } else {
      const int j = lb - g.nA, ups = upt / g.hs;
      u0 = (g.cus * g.hw + j / g.hs) * upt + (j % g.hs) * ups; u1 = u0 + ups;
    }
  } else {
    lb = ((G & 7) == 0) ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    // which blocks get one allotment more: the first `rem` logical blocks -- or (whole tiles, g.bal) the first rem / 8 blocks of
    // EVERY XCD, which the dispatcher puts on different CUs.  (Logical blocks are XCD-contiguous: with "the first rem" two of
    // the eight XCDs got all the extra tiles and a 1.25-round launch ran like a 2-round one.)
    int before = lb < g.rem ? lb : g.rem, mine = lb < g.rem ? 1 : 0;
    if (g.bal && (G & 7) == 0) {
      const int x = blockIdx.x & 7, i = blockIdx.x >> 3, r8 = g.rem >> 3, rx = g.rem & 7;
      const int lim = r8 + (x < rx ? 1 : 0);
      before = x * r8 + (x < rx ? x : rx) + (i < lim ? i : lim);
      mine = i < lim ? 1 : 0;
    }
    u0 = (lb * g.base + before) * g.gran;
    u1 = u0 + (g.base + mine) * g.gran;
  }
  if (u0 >= u1) return;
#if defined(X3_EXP_STAGGER)         // probe: blocks start in three phases, X3_EXP_STAGGER x 0.85 us apart (de-phases the tile ends -- and with
  for (int q = 0; q < (int)(blockIdx.x % 3) * X3_EXP_STAGGER; ++q) __builtin_amdgcn_s_sleep(32);   // them the chip-wide bursts of result stores)
#endif
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
  auto tile_origin = [&](int id, int& m0, int& n0) {
    const int r = id % tiles_mn;
    m0 = (r / g.tiles_n) * BM; n0 = (r % g.tiles_n) * BN;
  };

  // ---- loader: uniform base pointers (advance per slice) + per-lane BYTE offsets (change per tile) ----
  int ld_p = 0, ld_j = 0, ld_end = 0;
  unsigned a_off[2] = {0, 0}, w_off[2] = {0, 0};   // (a fixed bound: an array of dependent size makes the DMA builtin's arguments type-dependent,
                                       //  and the host pass then drops the whole kernel instantiation without a diagnostic)
  const char* a_base = nullptr;
  const char* w_base = nullptr;
  auto set_tile = [&](int id, int j0) {
    int m0, n0;
    tile_origin(id, m0, n0);
    const float* Asrc = (g.A2 != nullptr && n0 >= g.n_split) ? g.A2 : g.A;
    a_base = reinterpret_cast<const char*>(Asrc) + (size_t)j0 * (BK * 4);
    w_base = reinterpret_cast<const char*>(g.W) + (size_t)j0 * (BK * 4);
#pragma unroll
    for (int q = 0; q < NPA; ++q) {   // piece = 16 rows x 64 B; lane: row lane / 4, 16-byte slot lane % 4 (swizzled by (row >> 2) & 3)
      const int lr = 16 * (wave * NPA + q) + (lane >> 2);
      int row = m0 + lr;
      row = row < g.M ? row : g.M - 1;
      a_off[q] = ((unsigned)row * g.lda + 4 * ((lane & 3) ^ ((lr >> 2) & 3))) * 4;
    }
#pragma unroll
    for (int q = 0; q < NPW; ++q) {   // W pieces: the same 16 rows x 64 B form
      const int lr = 16 * (wave * NPW + q) + (lane >> 2);
      int row = n0 + lr;
      row = row < g.N ? row : g.N - 1;
      w_off[q] = ((unsigned)(g.w_row0 + row) * g.ldw + 4 * ((lane & 3) ^ ((lr >> 2) & 3))) * 4;
    }
  };
  auto issue_piece = [&](int k, int slot) {   // piece k of the slice the loader stands on -> ring slot `slot`
    unsigned char* base = lds + slot * SLOT;
#if defined(X3_EXP_NODMA)          // probe (see gemm_x3_kernel)
    if (g.M > 0) return;
#endif
    if (k < NPA) {
      __builtin_amdgcn_global_load_lds(a_base + a_off[k < NPA ? k : 0], X3_LDS_PTR(base + (wave * NPA + k) * 1024), 16, 0, 0);
    } else {
      const int q = k - NPA;
      __builtin_amdgcn_global_load_lds(w_base + w_off[q < NPW ? q : 0], X3_LDS_PTR(base + A_REG + (wave * NPW + (q < NPW ? q : 0)) * 1024), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++ld_j == ld_end) {
      if (ld_p + 1 < nseg) {
        int tile, j0, n, kind;
        segment(++ld_p, tile, j0, n, kind);
        set_tile(tile, j0);
        ld_j = j0; ld_end = j0 + n;
      } else {
        ld_j = ld_end - 1;   // past the end: the last slice again (never read)
      }
    } else { a_base += BK * 4; w_base += BK * 4; }
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int k = 0; k < NP; ++k) issue_piece(k, slot);
    advance();
  };

  // ---- fragment addresses (bytes inside a slot) ----
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned fw_r0 = A_REG + (wn * NI * 32 + l32) * 64 + (((2 * half) ^ ((l32 >> 2) & 3)) * 16);       // + ni * 2048
  const unsigned fw_r1 = A_REG + (wn * NI * 32 + l32) * 64 + (((2 * half + 1) ^ ((l32 >> 2) & 3)) * 16);
  const unsigned fa_r0 = (32 * wm + l32) * 64 + (((2 * half) ^ ((l32 >> 2) & 3)) * 16);         // floats k = 8 half .. + 3
  const unsigned fa_r1 = (32 * wm + l32) * 64 + (((2 * half + 1) ^ ((l32 >> 2) & 3)) * 16);

  u32x4 wf[2][2][NI];   // [set][16-byte half of the lane's 8 floats][ni]
  u32x4 af[2][2];       // [set][half]: the lane's 8 floats of its A row (MODE 1: normalised in place)
  f32x16 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
  auto read_w = [&](u32x4 (&dst)[2][NI], unsigned slot_addr, int r) {   // r = j * NI + ni, a constant after unrolling
    const int j = r / NI, ni = r % NI;
    if (ni == 0) dst[j][0] = x3_lds_read16<0>(slot_addr + (j ? fw_r1 : fw_r0));
    else dst[j][NI > 1 ? 1 : 0] = x3_lds_read16<2048>(slot_addr + (j ? fw_r1 : fw_r0));
  };

  // ---- MODE 1: (mean, rstd) of this lane's row for the tile being computed and the next one ----
  float mean_c = 0.f, rstd_c = 1.f, mean_n = 0.f, rstd_n = 1.f;
  float mean_s = 0.f, rstd_s = 1.f;   // what the split in flight uses
  const unsigned stat_lds = lds0 + RING * SLOT + wave * 4096;
  auto stats_fetch = [&](int p) {     // DMA the segment statistics of segment p's rows (this wave's 32) into the wave's patch
    if (MODE != 1 || p >= nseg) return;
    int tile, j0, n, kind, m0, n0;
    segment(p, tile, j0, n, kind);
    tile_origin(tile, m0, n0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {     // piece = 8 rows x 128 B
      int row = m0 + 32 * wm + 8 * q + (lane >> 3);
      row = row < g.M ? row : g.M - 1;
      // (16-byte chunk c of patch row r sits at position c ^ ((r >> 1) & 7): the merge's reads are then bank-conflict free)
      __builtin_amdgcn_global_load_lds(g.ln_in + (size_t)row * 32 + ((lane & 7) ^ ((4 * q + (lane >> 4)) & 7)) * 4,
                                       X3_LDS_PTR(lds + RING * SLOT + wave * 4096 + q * 1024), 16, 0, 0);
    }
  };
  auto stats_merge = [&](float& mean, float& rstd) {   // Chan's update for 16 equal parts (as ff_ln_finish of the f32 family)
    if (MODE != 1) return;
    // Row l32 of the patch, chunk c at position c ^ ((l32 >> 1) & 7) (see stats_fetch): the 16 lanes of a ds_read_b128 lane
    // group then touch 16 different 16-byte slots of the 256-byte bank row.  (Unswizzled, 128-byte row strides put 8 lanes of
    // a group on the same four banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.40 for this kernel form in round 5.)
    u32x4 s[8];
    const unsigned srow = stat_lds + l32 * 128, skx = ((l32 >> 1) & 7) << 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = x3_lds_read16<0>(srow + ((16u * c) ^ skx));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7])::"memory");
    float sm = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, s[i]);
      sm += v[0] + v[2]; m2 += v[1] + v[3];
    }
    mean = sm * (1.0f / 16.0f);
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = __builtin_bit_cast(f32x4, s[i]);
      const float d0 = v[0] - mean, d1 = v[2] - mean;
      dev += d0 * d0 + d1 * d1;
    }
    const float var = (m2 + 32.f * dev) * (1.0f / 512.0f);
    rstd = 1.0f / sqrtf(var + g.ln_eps);
  };

  // MODE 1: (x - mean) * rstd on float pair p (0..3) of the 8 floats in `dst` (the reference's order of operations)
  auto norm_pair = [&](u32x4 (&dst)[2], int p) {
    f32x4 v = __builtin_bit_cast(f32x4, dst[p >> 1]);
    const int c = 2 * (p & 1);
    v[c] = x3_fmul(x3_fsub(v[c], mean_s), rstd_s);
    v[c + 1] = x3_fmul(x3_fsub(v[c + 1], mean_s), rstd_s);
    dst[p >> 1] = __builtin_bit_cast(u32x4, v);
    asm volatile("" : "+v"(dst[p >> 1]));
  };

  // ---- compute-side segment state ----
  int cp_p = 0, cp_cnt = 0, cp_n = 0, cp_kind = 0;
  int e_m0 = 0, e_n0 = 0;
  auto begin_segment = [&](int p) {
    int id, j0;
    segment(p, id, j0, cp_n, cp_kind);
    cp_cnt = 0;
    tile_origin(id, e_m0, e_n0);
  };

  // End of a segment: hand the partial tile over (kind 1), or finish the tile -- after adding the partials of the
  // lower-numbered blocks (kind 2) -- with bias / activation / table / residual, the store and (MODE 2) the statistics.
  // Every global access of this path is inline assembly with its own waits: accesses the compiler knows about leave an
  // "unknown" wait-count state at the join with the K loop (it then opens every slice with s_waitcnt vmcnt(0)), and the
  // stores must not be waited for at all -- they drain behind the next tile's first slices.
  auto gload16 = [&](const float* ptr) -> f32x4 {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
  };
  auto end_segment = [&]() {
    constexpr int NQ = 4 * NI;   // 16-byte accumulator groups per lane
    if (cp_kind == 1) {
      f32x4* wp = reinterpret_cast<f32x4*>(g.ws + (size_t)lb * (BM * BN)) + tid;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[ni][4 * q], acc[ni][4 * q + 1], acc[ni][4 * q + 2], acc[ni][4 * q + 3]};
          asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(wp + (ni * 4 + q) * 256), "v"(v) : "memory");
          acc[ni][4 * q] = 0.f; acc[ni][4 * q + 1] = 0.f; acc[ni][4 * q + 2] = 0.f; acc[ni][4 * q + 3] = 0.f;
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (tid == 0) __hip_atomic_store(g.flags + lb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (cp_kind == 2) {  // add the partials of the blocks that hold units [k0 * upt, u0) of this tile
      int c0;
      if (g.hyb) {
        c0 = lb - (g.hs - 1);               // the other pieces of this tile: the hs - 1 block ids below this one
      } else {
        const int ab = (k0 * upt) / g.gran;   // allotment that starts the tile (unit ranges: gran = 1)
        const int big = g.rem * (g.base + 1);
        c0 = ab < big ? ab / (g.base + 1) : g.rem + (ab - big) / g.base;
      }
      for (int c = c0; c < lb; ++c) {
        while (__hip_atomic_load(g.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const f32x4* rp = reinterpret_cast<const f32x4*>(g.ws + (size_t)c * (BM * BN)) + tid;
        constexpr int TQ = NQ < 8 ? NQ : 8;   // 8 x 16 bytes in flight per lane and round trip (all 16 of a 128-row tile would spill)
#pragma unroll
        for (int h = 0; h < NQ / TQ; ++h) {
          f32x4 t[TQ];
#pragma unroll
          for (int j = 0; j < TQ; ++j)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[j]) : "v"(rp + (h * TQ + j) * 256) : "memory");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int j = 0; j < TQ; ++j) {
            asm volatile("" : "+v"(t[j]));
            const int q = h * TQ + j;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[q >> 2][4 * (q & 3) + i] += t[j][i];
          }
        }
      }
      __builtin_amdgcn_s_barrier();  // every thread is past its flag polls
      if (tid < lb - c0) __hip_atomic_store(g.flags + c0 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // This lane: row m, columns n0 + (wn * NI + ni) * 32 + 8 q + 4 half + {0..3}.  ALL epilogue operands of the tile are
    // requested in one go (the fragments of the next slice are not live here: they are read again behind the epilogue),
    // every result is finished in its accumulator registers, then the stores go out back to back.  A lane reads and
    // writes the same elements, so a residual that aliases C stays correct.
    const int m = e_m0 + 32 * wm + l32;
    const bool rowok = m < g.M;
    const int mc = rowok ? m : g.M - 1;
    const int nb0 = e_n0 + wn * NI * 32 + 4 * half;
    const bool has_tab = (MODE == 1 || MODE == 3) && g.rowtab != nullptr;
    const float* xrow = nullptr;   // residual row, or (MODE 1) the row of the position table
    int xlim = 0;
    if (g.res) { xrow = g.res + (size_t)mc * g.ldr; xlim = g.N; }
    else if (has_tab) { xrow = g.rowtab + (size_t)(mc / g.rowtab_div) * g.ld_rowtab; xlim = g.rowtab_cols; }
    // MODE 3: LayerNorm(x) W'^T = rstd (x W'^T - mean colsum(W')): the K loop multiplied the RAW rows; this row's statistics are
    // merged here from the producer's 16 segment statistics (Chan's update, as stats_merge) and applied to the finished sums.
    float ln_mean = 0.f, ln_rstd = 1.f;
    if (MODE == 3) {
      f32x4 sv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sv[i] = gload16(g.ln_in + (size_t)mc * 32 + 4 * i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(sv[i]));
      float sm = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { sm += sv[i][0] + sv[i][2]; m2 += sv[i][1] + sv[i][3]; }
      ln_mean = sm * (1.0f / 16.0f);
      float dev = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d0 = sv[i][0] - ln_mean, d1 = sv[i][2] - ln_mean; dev += d0 * d0 + d1 * d1; }
      ln_rstd = 1.0f / sqrtf((m2 + 32.f * dev) * (1.0f / 512.0f) + g.ln_eps);
    }
    // a few 32-column groups at a time: with all of a 128-column row's operands in flight at once the register allocator
    // spills loop-invariant addresses INTO the K loop
    constexpr int GQ = (NI == 4 || MODE == 3) ? 4 : (NQ < 8 ? NQ : 8);
#pragma unroll
    for (int h = 0; h < NQ / GQ; ++h) {
      f32x4 xv[GQ], bv[GQ], cv[GQ];
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        const int q = h * GQ + j;
        const int n = nb0 + (q >> 2) * 32 + (q & 3) * 8;
        xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        cv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if !defined(X3_EXP_NOEPILOAD)      // probe: the epilogue without its bias / residual / table loads (wrong numbers)
        if (xrow && n + 3 < xlim) xv[j] = gload16(xrow + n);
        if (g.bias && n + 3 < g.N) bv[j] = gload16(g.bias + n);
#endif
        if (MODE == 3 && n + 3 < g.N) cv[j] = gload16(g.colsum + g.w_row0 + n);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        asm volatile("" : "+v"(xv[j]), "+v"(bv[j]));
        if (MODE == 3) asm volatile("" : "+v"(cv[j]));
      }
#pragma unroll
      for (int j = 0; j < GQ; ++j) {
        const int q = h * GQ + j, ni = q >> 2, qq = q & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = acc[ni][4 * qq + i];
          if (MODE == 3) v = (v - ln_mean * cv[j][i]) * ln_rstd;
          v += bv[j][i];
          if (MODE == 1 || MODE == 3) {   // position-table term in front of the activation, no residual
            v += xv[j][i];
            if (g.act == 1) v = fmaxf(v, 0.f);
          } else {
            if (g.act == 1) v = fmaxf(v, 0.f);
            v += xv[j][i];
          }
          acc[ni][4 * qq + i] = v;
        }
      }
    }
    if (MODE == 2) {   // (mean, M2) of this row's 32 stored values per column group: 16 here, 16 in the lane 32 away
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) sm += acc[ni][e];
        sm = ff_halves_sum(sm);
        const float mean = sm * (1.0f / 32.0f);
        float m2 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = acc[ni][e] - mean; m2 += d * d; }
        m2 = ff_halves_sum(m2);
        const int seg = (e_n0 >> 5) + wn * NI + ni;
        if (half == 0 && rowok && seg * 32 < g.N) {
          const f32x2 st2 = f32x2{mean, m2};
          asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(g.ln_out + ((size_t)m * (g.N >> 5) + seg) * 2), "v"(st2) : "memory");
        }
      }
    }
    if (rowok) {
      float* cp = g.C + (size_t)m * g.ldc + nb0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (nb0 + (q >> 2) * 32 + (q & 3) * 8 + 3 < g.N) {
          const f32x4 v = {acc[q >> 2][4 * (q & 3)], acc[q >> 2][4 * (q & 3) + 1], acc[q >> 2][4 * (q & 3) + 2], acc[q >> 2][4 * (q & 3) + 3]};
#if defined(X3_EXP_NOSTORE)        // probe: results are not stored (one store per lane and tile keeps the accumulators alive)
          if (q == 0)
#endif
#if defined(X3_EXP_NTSTORE)        // probe: non-temporal result stores
          asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(cp + (q >> 2) * 32 + (q & 3) * 8), "v"(v) : "memory");
#else
          asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(cp + (q >> 2) * 32 + (q & 3) * 8), "v"(v) : "memory");
#endif
        }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
  };

  // ---- prologue: slices 0, 1, 2 in flight; 0 and 1 landed; fragments of slice 0 in set 0 ----
  {
    int tile, j0, n, kind;
    segment(0, tile, j0, n, kind);
    set_tile(tile, j0);
    ld_j = j0; ld_end = j0 + n;
  }
  if (MODE == 1) {   // statistics of segment 0 (and 1): fetched and merged before the first split
    stats_fetch(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stats_merge(mean_c, rstd_c);
    stats_fetch(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    stats_merge(mean_n, rstd_n);
    stats_fetch(2);
    mean_s = mean_c; rstd_s = rstd_c;
  }
#pragma unroll
  for (int r = 0; r < RING; ++r) issue(r);
  begin_segment(0);
  static_assert((RING - 2) * NP <= 15, "vmcnt immediate");
  __builtin_amdgcn_s_waitcnt(0x0F70 | ((RING - 2) * NP));   // vmcnt: slices 0 and 1 have landed
  __builtin_amdgcn_s_barrier();
  {
    af[0][0] = x3_lds_read16(lds0 + fa_r0);
    af[0][1] = x3_lds_read16(lds0 + fa_r1);
#pragma unroll
    for (int r = 0; r < 2 * NI; ++r) read_w(wf[0], lds0, r);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1])::"memory");
    if (MODE == 1) {
#pragma unroll
      for (int p = 0; p < 4; ++p) norm_pair(af[0], p);
    }
  }
  __builtin_amdgcn_s_barrier();

  // where the normalisation (MODE 1) and the DMA pieces go among the MFMA gaps
  constexpr int SP0 = 3;              // the two A reads are the oldest of the 4 reads issued by then
  constexpr int DM0 = NRD + 2;        // first DMA piece
  int s0 = 0, s1 = 1;   // ring slots of slice s, s + 1  (slice s + 3 goes to slot s0)
  const int total = 2 * (u1 - u0);
  for (int s = 0; s < total; s += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned nb = lds0 + s1 * SLOT;   // slot of slice s + 1
      if (MODE == 1) {   // the rows split in this iteration belong to the next segment's tile when this is the segment's last slice
        const bool nx = cp_cnt + 1 == cp_n;
        mean_s = nx ? mean_n : mean_c; rstd_s = nx ? rstd_n : rstd_c;
      }
      // NMF MFMAs of slice s; in their gaps: the reads of slice s + 1 (one per gap), (MODE 1) its normalisation, the DMA of s + 3
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        const int e = i / NI, ni = i % NI;
#if defined(X3_EXP_NOMFMA)
        if (g.M < 0)
#endif
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(f32x4, wf[u][e >> 2][ni])[e & 3],
                                                       __builtin_bit_cast(f32x4, af[u][e >> 2])[e & 3], acc[ni], 0, 0, 0);
        if (i == 0) af[u ^ 1][0] = x3_lds_read16(nb + fa_r0);
        else if (i == 1) af[u ^ 1][1] = x3_lds_read16(nb + fa_r1);
        else if (i < NRD) read_w(wf[u ^ 1], nb, i - 2);
        if (MODE == 1) {
          if (i == SP0) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[u ^ 1][0]), "+v"(af[u ^ 1][1])::"memory");
          if (i >= SP0 && i < SP0 + 4) norm_pair(af[u ^ 1], i - SP0);
        }
        if (i >= DM0 && i < DM0 + NP) issue_piece(i - DM0, s0);
        if (i == NMF - 1) advance();
        __builtin_amdgcn_sched_barrier(0);
      }
      // slice s + 2 has landed (own pieces), every fragment of slice s + 1 is in registers
      __builtin_amdgcn_s_waitcnt(0x0070 | ((RING - 2) * NP));   // vmcnt: everything but the youngest RING - 2 slices; lgkmcnt(0)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("" : "+v"(af[u ^ 1][j]));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[u ^ 1][j][ni]));
      }
      if (++cp_cnt == cp_n) {  // block-uniform: the last slice of the segment was just issued
        end_segment();
        if (++cp_p < nseg) begin_segment(cp_p);
        // 64-row tiles: the fragments of slice s + 1 are read AGAIN here (their slot is untouched until the barrier below),
        // so the values read inside the loop are dead across the epilogue and its operands do not compete with them for the
        // 168 registers of three waves per SIMD.  (128-row tiles keep them live: the second copy of the read-and-split code
        // costs that kernel more registers than it frees.)  Either way no register that an asynchronous ds_read has not
        // filled yet may be spilled: tools/check_x3_asm.py looks for scratch accesses inside the MFMA runs of every build.
        if (NI == 2) {
          af[u ^ 1][0] = x3_lds_read16(nb + fa_r0);
          af[u ^ 1][1] = x3_lds_read16(nb + fa_r1);
#pragma unroll
          for (int r = 0; r < 2 * NI; ++r) read_w(wf[u ^ 1], nb, r);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[u ^ 1][0]), "+v"(af[u ^ 1][1])::"memory");
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" : "+v"(wf[u ^ 1][j][ni]));
          mean_s = mean_n; rstd_s = rstd_n;   // (MODE 1: these rows belong to the segment that starts now)
          if (MODE == 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) norm_pair(af[u ^ 1], p);
          }
        }
        if (MODE == 1) {       // statistics: the next segment's become current, the one after is merged from the patch
          mean_c = mean_n; rstd_c = rstd_n;
          stats_merge(mean_n, rstd_n);
          stats_fetch(cp_p + 2);
        }
      }
      __builtin_amdgcn_s_barrier();
      s0 = s1; s1 = s1 + 1 == RING ? 0 : s1 + 1;
    }
  }
}

// Partial-tile workspace: one per (device, stream), as in ff_gemm.hip.
constexpr int X3_MAX_GRID = 768;
constexpr size_t X3_WS_BYTES = (size_t)X3_MAX_GRID * 64 * 128 * sizeof(float);   // one 32 KB partial tile per block
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
  FF_CHECK_HIP(hipMalloc(&w.ws, X3_WS_BYTES));
  FF_CHECK_HIP(hipMalloc(&w.flags, X3_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipMemset(w.flags, 0, X3_MAX_GRID * sizeof(unsigned int)));
  FF_CHECK_HIP(hipDeviceSynchronize());
  g_x3.push_back(w);
  out->ws = w.ws; out->flags = w.flags;
  return FF_OK;
}

// tuning / tests: force the launch shape (0 auto, 1 whole tiles, 2 unit ranges)
int g_x3_force_shape = 0;

template <int BM, int MODE, int NT = 3>
int x3_launch_mode(const X3Args& g, int grid, hipStream_t st) {
  static std::atomic<bool> attr_set[16] = {};   // hipFuncSetAttribute is per device; host threads may race here (idempotent)
#ifndef X3_EXP_LDS_PAD     // probe: extra dynamic LDS per block = fewer blocks per CU (occupancy experiments)
#define X3_EXP_LDS_PAD 0
#endif
  constexpr int bytes = x3_ring<MODE, NT>() * (BM * 64 + NT * X3_BN * 32) + (MODE == 1 ? X3_STAT_BYTES : 0) + X3_EXP_LDS_PAD;
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<BM, MODE, NT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_x3_kernel<BM, MODE, NT>), dim3(grid), dim3(256), bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
template <int BM, int MODE, int BN = X3_BN>
int dma_f32_launch_mode(const X3Args& g, int grid, hipStream_t st) {
  static std::atomic<bool> attr_set[16] = {};
  constexpr int bytes = X3_RING_F32 * (BM * 64 + BN * 64) + (MODE == 1 ? X3_STAT_BYTES : 0);
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dma_f32_kernel<BM, MODE, BN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((gemm_dma_f32_kernel<BM, MODE, BN>), dim3(grid), dim3(256), bytes, st, g);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// Launch shape: WHOLE TILES, always (round 4).  Measured over the decode's shapes (profiles/r04/gemm_x3_variants.txt): once the
// blocks that get one tile more are dealt out over the XCDs and CUs, whole tiles are as fast as or faster than equal K-unit
// ranges everywhere but one shape (9216 x 1024 -> 512: 152 vs 163 TF/s) -- the exchange of the cut tiles (a 32 KB hand-over
// and a fix-up round trip per block) costs what the finer balance buys -- and they need no cross-block traffic and sum every
// output element in ONE fixed order whatever the grid.  The unit-range form stays available (ff_set_x3_tuning: tests, probes).
// Only the 64-row tile is instantiated: the 128-row form (4 x 1 waves of 32 x 128) needs all 256 registers of a two-wave SIMD
// and the register allocator then spills INTO the K loop (a spill of a fragment register that an asynchronous ds_read has
// not filled yet stores garbage, and every reload costs a full vmcnt drain); where it compiled cleanly it was within +-7 %
// of the 64-row tile.
int x3_launch(X3Args g, int mode, hipStream_t st, bool f32 = false, int bn = X3_BN) {
  const int M = g.M, N = g.N, K = g.K;
  constexpr int BM = 64;
  const int cus = ff_num_cus();   // 256 on an MI355X in SPX mode; a partition (CPX / fewer CUs) gets its own launch shape
  // block slots per CU: registers / the statistics patch of the LayerNorm consumers decide
  // (the 64-column f32 tile needs 109-111 registers without the LayerNorm consumer's state: four blocks fit a CU)
  const int spc = f32 ? ((bn == 64 && mode != 1) ? 4 : 3) : (mode == 3 ? 2 : (mode == 1 ? (g.nt == 2 ? X3_LN_H_BLOCKS : 2) : 3));
  const int slots = cus * spc;
  g.tiles_n = ff_cdiv(N, bn);
  g.tiles_m = ff_cdiv(M, BM);
  g.upt = K / 32;
  const long tiles = (long)g.tiles_m * g.tiles_n;
  const long units = tiles * g.upt;
  FF_CHECK_ARG(units < (1L << 30), "ff_gemm_x3: problem too large");
  int shape = g_x3_force_shape ? g_x3_force_shape : 1;
  if (units < 2) shape = 1;
  long grid;
  // Hybrid (the default whenever it applies): hw = tiles / CUs whole tiles per CU and the r = tiles % CUs remaining tiles cut
  // into hs K-pieces so that r * hs <= CUs.  Measured on the decode's N = 512 projections a launch of 2.06 tiles per CU then
  // costs ~2.2 tile times instead of 3 (profiles/r04/gemm_x3_hybrid.txt).
  const long hw = tiles / cus, hr = tiles % cus;
  int hs = 0;
  for (int c = 8; c >= 2; c >>= 1)
    if (hr * c <= cus && g.upt % c == 0 && g.upt / c >= 2) { hs = c; break; }
  const int small_split = ff_knob(FF_K_X3_SMALL_SPLIT);   // (probe: K-pieces for launches below one tile per CU)
  if (!g_x3_force_shape && ((hw >= 1 && hr > 0) || (small_split && hw == 0 && hr * 2 <= cus)) && hs > 0) {
    g.hyb = 1; g.hw = (int)hw; g.hs = hs; g.cus = cus;
    g.ha = (spc == 2 || hw == 1) ? 1 : 2;   // (one slot stays for the K-piece blocks)
    g.nA = hw == 0 ? 0 : cus * g.ha;
    grid = g.nA + hr * hs;
  } else if (shape == 1) {        // whole tiles: contiguous runs of tiles per block
    grid = tiles < slots ? tiles : slots;
    g.gran = g.upt;
    g.base = (int)(tiles / grid);
    g.rem = (int)(tiles % grid);
    g.bal = 1;
  } else {                 // equal unit ranges over every resident block slot
    grid = units < slots ? units : slots;
    g.gran = 1;
    g.base = (int)(units / grid);
    g.rem = (int)(units % grid);
  }
  FF_RETURN_IF(x3_acquire(st, &g));
  if (f32 && bn == 64) {
    if (mode == 1) return dma_f32_launch_mode<BM, 1, 64>(g, (int)grid, st);
    if (mode == 2) return dma_f32_launch_mode<BM, 2, 64>(g, (int)grid, st);
    return dma_f32_launch_mode<BM, 0, 64>(g, (int)grid, st);
  }
  if (f32) {   // (the caller opened the f32 family's profiling scope)
    if (mode == 1) return dma_f32_launch_mode<BM, 1>(g, (int)grid, st);
    if (mode == 2) return dma_f32_launch_mode<BM, 2>(g, (int)grid, st);
    return dma_f32_launch_mode<BM, 0>(g, (int)grid, st);
  }
  FFProfScope prof(FF_CAT_GEMM_X3, 2.0 * M * N * K, st);
  ff_prof_add_bytes(FF_CAT_GEMM_X3, 4.0 * (double)M * K + 2.0 * (g.nt == 2 ? 2 : 3) * (double)N * K + 4.0 * (double)M * N * (g.res ? 2 : 1));
  if (g.nt == 2) {
    if (mode == 1) return x3_launch_mode<BM, 1, 2>(g, (int)grid, st);
    if (mode == 2) return x3_launch_mode<BM, 2, 2>(g, (int)grid, st);
    if (mode == 3) return x3_launch_mode<BM, 3, 2>(g, (int)grid, st);
    return x3_launch_mode<BM, 0, 2>(g, (int)grid, st);
  }
  if (mode == 1) return x3_launch_mode<BM, 1>(g, (int)grid, st);
  if (mode == 2) return x3_launch_mode<BM, 2>(g, (int)grid, st);
  if (mode == 3) return x3_launch_mode<BM, 3>(g, (int)grid, st);
  return x3_launch_mode<BM, 0>(g, (int)grid, st);
}

int x3_check_common(const float* A, int lda, const void* w_planes, const float* bias, const float* residual, int ldr,
                    float* C, int ldc, int M, int N, int K, int act, const char* who) {
  FF_CHECK_ARG(M > 0 && N > 0 && K >= 64 && (K % 32) == 0, "%s: bad M=%d N=%d K=%d (K %% 32, K >= 64)", who, M, N, K);
  FF_CHECK_ARG(A && w_planes && C, "%s: null operand", who);
  FF_CHECK_ARG(ff_aligned16(w_planes), "%s: weight planes must be 16-byte aligned", who);
  FF_CHECK_ARG((lda & 3) == 0 && lda >= K && ff_aligned16(A), "%s: A must be 16-byte aligned with lda %% 4 == 0 (lda=%d)", who, lda);
  FF_CHECK_ARG((size_t)M * lda < ((size_t)1 << 30), "%s: A is too large for 32-bit byte offsets", who);
  FF_CHECK_ARG((N & 3) == 0 && N >= 4 && ldc >= N && (ldc & 3) == 0 && ff_aligned16(C),
               "%s: N %% 4, ldc %% 4 and a 16-byte aligned C are required (N=%d ldc=%d)", who, N, ldc);
  FF_CHECK_ARG(!bias || ff_aligned16(bias), "%s: bias must be 16-byte aligned", who);
  FF_CHECK_ARG(!residual || (ldr >= N && (ldr & 3) == 0 && ff_aligned16(residual)), "%s: bad residual (ldr %% 4, 16-byte aligned)", who);
  FF_CHECK_ARG(act == 0 || act == 1, "%s: act must be 0 or 1", who);
  return FF_OK;
}

}  // namespace

// ---- f32 family: the dispatcher of ff_gemm.hip hands eligible launches to gemm_dma_f32_kernel ----
bool ff_gemm_dma_f32_ok(const GemmArgs& a, int batch) {
  if (batch != 1 || a.K < 64 || (a.K % 32) != 0 || (a.N & 3) != 0 || a.N < 4) return false;
  if ((a.lda & 3) || (a.ldw & 3) || (a.ldc & 3) || (a.res && (a.ldr & 3))) return false;
  if (!ff_aligned16(a.A) || !ff_aligned16(a.W) || !ff_aligned16(a.C) || (a.A2 && !ff_aligned16(a.A2)) ||
      (a.bias && !ff_aligned16(a.bias)) || (a.res && !ff_aligned16(a.res))) return false;
  if (a.A2 && (a.n_split % 128) != 0) return false;
  if ((size_t)a.M * a.lda >= ((size_t)1 << 30) || (size_t)a.N * a.ldw >= ((size_t)1 << 30)) return false;
  if (a.ln_in && (a.K != 512 || a.ln_nseg != 16 || !ff_aligned16(a.ln_in))) return false;
  if (a.rowtab && ((a.rowtab_cols & 3) || (a.ld_rowtab & 3) || !ff_aligned16(a.rowtab) || a.res)) return false;
  if (a.ln_out && (a.N & 31)) return false;
  return true;
}
int ff_gemm_dma_f32(const GemmArgs& a, hipStream_t st, int bn) {
  X3Args g;
  memset(&g, 0, sizeof(g));
  g.A = a.A; g.A2 = a.A2; g.lda = a.lda;
  g.W = a.W; g.ldw = a.ldw; g.bias = a.bias; g.res = a.res; g.ldr = a.ldr;
  g.C = a.C; g.ldc = a.ldc;
  g.M = a.M; g.N = a.N; g.K = a.K; g.n_split = a.A2 ? a.n_split : a.N; g.act = a.act;
  g.w_rows = a.N; g.w_row0 = 0;
  g.ln_in = a.ln_in; g.ln_eps = a.ln_eps;
  g.rowtab = a.rowtab; g.ld_rowtab = a.ld_rowtab; g.rowtab_div = a.rowtab_div > 0 ? a.rowtab_div : 1; g.rowtab_cols = a.rowtab_cols;
  g.ln_out = a.ln_out;
  return x3_launch(g, a.ln_in ? 1 : (a.ln_out ? 2 : 0), st, true, bn == 64 ? 64 : X3_BN);
}

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

extern "C" size_t ff_split_weight_fp16x2_bytes(int N, int K) { return (size_t)2 * N * K * sizeof(unsigned short); }

extern "C" int ff_split_weight_fp16x2(const float* W, int ldw, int N, int K, void* planes, ff_stream_t stream) {
  FF_CHECK_ARG(W && planes && N > 0 && K > 0 && (K & 15) == 0 && ldw >= K, "ff_split_weight_fp16x2: bad arguments (K %% 16)");
  FF_CHECK_ARG(ff_aligned16(planes), "ff_split_weight_fp16x2: planes must be 16-byte aligned");
  const size_t n2 = (size_t)N * (K / 2);
  const int grid = (int)((n2 + 255) / 256 < 4096 ? (n2 + 255) / 256 : 4096);
  hipLaunchKernelGGL(split_weight_fp16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, W, ldw, N, K,
                     static_cast<unsigned short*>(planes));
  FF_CHECK_LAUNCH();
  return FF_OK;
}

extern "C" int ff_x3_prepare_stream(hipStream_t st) {
  X3Args g;
  return x3_acquire(st, &g);
}

// The same per-(device, stream) area as scratch memory of other kernels on that stream (stream order keeps the users apart):
// the K/V-resident attention kernel parks its first partial records there (ff_attention.hip).
int ff_stream_scratch(hipStream_t st, size_t bytes, float** out) {
  FF_CHECK_ARG(bytes <= X3_WS_BYTES, "ff_stream_scratch: %zu bytes requested, the area has %zu", bytes, X3_WS_BYTES);
  X3Args g;
  FF_RETURN_IF(x3_acquire(st, &g));
  *out = g.ws;
  return FF_OK;
}

extern "C" int ff_set_x3_tuning(int shape) {
  FF_CHECK_ARG(shape >= 0 && shape <= 2, "ff_set_x3_tuning: shape in {0, 1, 2}");
  g_x3_force_shape = shape;
  ff_tuning_changed();
  return FF_OK;
}

namespace {
int gemm_split_plain(int nt, const char* who, const float* A, int lda, const float* A2, int n_split, const void* w_planes,
                     const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N,
                     int K, int act, ff_stream_t stream) {
  if (M == 0 || N == 0) return FF_OK;
  FF_RETURN_IF(x3_check_common(A, lda, w_planes, bias, residual, ldr, C, ldc, M, N, K, act, who));
  FF_CHECK_ARG(!A2 || ff_aligned16(A2), "%s: A2 must be 16-byte aligned", who);
  if (A2) FF_CHECK_ARG(n_split > 0 && n_split < N && (n_split % 128) == 0, "%s: n_split must be a multiple of 128 inside (0,N)", who);
  X3Args g;
  memset(&g, 0, sizeof(g));
  g.nt = nt;
  g.A = A; g.A2 = A2; g.lda = lda;
  g.Wp = static_cast<const unsigned short*>(w_planes); g.bias = bias; g.res = residual; g.ldr = ldr;
  g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.n_split = A2 ? n_split : N; g.act = act;
  g.plane_stride = (long long)N * K;
  g.w_rows = N; g.w_row0 = 0;
  return x3_launch(g, 0, (hipStream_t)stream);
}
}  // namespace

extern "C" int ff_gemm_x3(const float* A, int lda, const float* A2, int n_split, const void* w_planes,
                          const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N,
                          int K, int act, ff_stream_t stream) {
  return gemm_split_plain(3, "ff_gemm_x3", A, lda, A2, n_split, w_planes, bias, residual, ldr, C, ldc, M, N, K, act, stream);
}
extern "C" int ff_gemm_x2h(const float* A, int lda, const float* A2, int n_split, const void* w_planes,
                           const float* bias, const float* residual, int ldr, float* C, int ldc, int M, int N,
                           int K, int act, ff_stream_t stream) {
  return gemm_split_plain(2, "ff_gemm_x2h", A, lda, A2, n_split, w_planes, bias, residual, ldr, C, ldc, M, N, K, act, stream);
}

// x3_ln_linear -- why the consumer's normalisation may move into the epilogue.  LN(x) W'^T = rstd (x W'^T - mean s), s = the row
// sums of W': the K loop then is the PLAIN loop (no statistics patch in LDS, no normalising VALU work), and the row statistics
// are merged where registers are free.  The price is numerical: the rounding error of the product is that of x W'^T, not of
// ((x - mean) rstd) W'^T, i.e. it grows by the factor (1 + |mean| / sigma) of the ROW.  On this model's LayerNorm inputs
// |mean| / sigma is 0.06 (median) / 0.15 (maximum over 50 836 rows of a gain-4 decode) and the two forms are equally far from
// fp64 (5.2e-7 vs 5.1e-7 relative); callers whose rows can have |mean| >> sigma pass w_colsum = NULL and get the
// normalise-first form (MODE 1).
namespace {
int gemm_split_ln(int nt, const ff_gemm_ln_desc* d, const void* w_planes, int plane_rows, int row0, const float* w_colsum,
                  ff_stream_t stream) {
  FF_CHECK_ARG(d != nullptr, "ff_gemm_x3_ln: null descriptor");
  const int M = d->M, N = d->N, K = d->K;
  if (M == 0 || N == 0) return FF_OK;
  if (plane_rows <= 0) plane_rows = N;
  FF_CHECK_ARG(row0 >= 0 && row0 + N <= plane_rows, "ff_gemm_x3_ln: rows [%d, %d) outside the %d rows of the planes", row0, row0 + N, plane_rows);
  FF_RETURN_IF(x3_check_common(d->A, d->lda, w_planes, d->bias, d->residual, d->ldr, d->C, d->ldc, M, N, K, d->act, "ff_gemm_x3_ln"));
  FF_CHECK_ARG(!(d->ln_stats_in && d->ln_stats_out), "ff_gemm_x3_ln: statistics in AND out in one launch are not supported");
  FF_CHECK_ARG(!d->ln_stats_in || (K == 512 && d->ln_nseg == 16 && d->ln_eps >= 0.f && ff_aligned16(d->ln_stats_in)),
               "ff_gemm_x3_ln: ln_stats_in needs K = 512 (16 segments of 32 columns), 16-byte aligned statistics");
  FF_CHECK_ARG(!d->row_table || (d->ln_stats_in && !d->residual && d->row_div > 0 && d->row_cols > 0 && d->row_cols <= N &&
                                 (d->row_cols & 3) == 0 && d->ld_row_table >= d->row_cols && (d->ld_row_table & 3) == 0 &&
                                 ff_aligned16(d->row_table)),
               "ff_gemm_x3_ln: row_table needs ln_stats_in, no residual, row_div > 0, 0 < row_cols <= N, row_cols / ld %% 4");
  FF_CHECK_ARG(!d->ln_stats_out || (N & 31) == 0, "ff_gemm_x3_ln: ln_stats_out needs N %% 32 == 0");
  X3Args g;
  memset(&g, 0, sizeof(g));
  g.nt = nt;
  g.A = d->A; g.lda = d->lda;
  g.Wp = static_cast<const unsigned short*>(w_planes); g.bias = d->bias; g.res = d->residual; g.ldr = d->ldr;
  g.C = d->C; g.ldc = d->ldc;
  g.M = M; g.N = N; g.K = K; g.n_split = N; g.act = d->act;
  g.plane_stride = (long long)plane_rows * K;
  g.w_rows = plane_rows; g.w_row0 = row0;
  g.ln_in = d->ln_stats_in; g.ln_eps = d->ln_eps;
  g.rowtab = d->row_table; g.ld_rowtab = d->ld_row_table; g.rowtab_div = d->row_div > 0 ? d->row_div : 1; g.rowtab_cols = d->row_cols;
  g.ln_out = d->ln_stats_out;
  g.colsum = w_colsum;
  FF_CHECK_ARG(!w_colsum || (ff_aligned16(w_colsum) && (row0 & 3) == 0), "ff_gemm_x3_ln: w_colsum must be 16-byte aligned, row0 %% 4 == 0");
  return x3_launch(g, d->ln_stats_in ? (w_colsum ? 3 : 1) : (d->ln_stats_out ? 2 : 0), (hipStream_t)stream);
}
}  // namespace

extern "C" int ff_gemm_x3_ln(const ff_gemm_ln_desc* d, const void* w_planes, int plane_rows, int row0, const float* w_colsum,
                             ff_stream_t stream) {
  return gemm_split_ln(3, d, w_planes, plane_rows, row0, w_colsum, stream);
}
// ... on fp16 planes (ff_split_weight_fp16x2).  w_colsum selects the epilogue form of the LayerNorm, which multiplies the RAW rows:
// the caller must know them to be inside fp16's range (|x| < 65504); the normalise-first form (w_colsum = NULL) needs no such bound.
extern "C" int ff_gemm_x2h_ln(const ff_gemm_ln_desc* d, const void* w_planes, int plane_rows, int row0, const float* w_colsum,
                              ff_stream_t stream) {
  return gemm_split_ln(2, d, w_planes, plane_rows, row0, w_colsum, stream);
}
