// Chain kernel: many operators of a decode step inside ONE persistent launch.
//
// The first decode steps of the parallel model, every step of the single-sequence model and the last-layer / last-row
// tail + pointer head of EVERY step are chains of 15..55 small dependent operators (a few hundred rows each): as
// separate launches each costs 6..15 us for < 1 us of matrix work -- dispatch, a cold first touch of its operands, a
// fill and a drain.  Here a fixed grid of co-resident workgroups (one per CU, 8 waves) interprets a list of operator
// descriptors; consecutive dependent operators are separated by a grid-wide phase boundary instead of a kernel boundary:
//
//   * operators  = the SAME device code as the stand-alone kernels (ff_device.h), compiled with COH = true: everything
//                  another workgroup of this launch may have produced is stored write-through (sc1) and loaded with
//                  agent-scope (sc1) loads, so a boundary needs no cache write-back / invalidate of its own;
//   * boundary   = every wave drains its stores (s_waitcnt vmcnt(0)), block barrier, ONE lane publishes the block's epoch
//                  flag (own 64-byte line); workgroup 0 gathers the G flags with G lanes in parallel and publishes a
//                  release word; one lane per workgroup polls it (relaxed agent loads + s_sleep).  Measured on MI355X
//                  (tools/ubench/grid_phase.hip): 1.5 us for 256 workgroups, 2.2-2.7 us including the first dependent
//                  sc1 load round trip -- against 3.6 us for a single contended atomic counter and 5-8 us for a kernel
//                  boundary with a cold first touch;
//   * residency  = the grid never exceeds the CU count and nothing else is queued on the stream, so every workgroup is
//                  resident; every poll loop is bounded, a timeout poisons the release word (later boundaries fall
//                  through) and is reported by ff_decode as an error instead of hanging the device.
#ifdef FF_EXPERIMENTAL   // (default builds carry none of this: ff_chain.h)
#include <mutex>
#include <vector>

#include "ff_common.h"
#include "ff_device.h"
#include "ff_chain.h"

namespace {

constexpr int CHAIN_WAVES = 8;
constexpr int CHAIN_THREADS = 64 * CHAIN_WAVES;
constexpr unsigned CHAIN_POISON = 0xFFFFFFFFu;
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 21;

__host__ __device__ constexpr int chain_lds_floats() {
  return ff_attention_wave_lds_floats(CHAIN_WAVES) > ff_gemm_small_lds_floats(2, CHAIN_WAVES) + 64
             ? ff_attention_wave_lds_floats(CHAIN_WAVES)
             : ff_gemm_small_lds_floats(2, CHAIN_WAVES) + 64;
}

struct ChainSync {
  unsigned* flags;     // [G][16] epoch of workgroup b at flags[16 b]; release word at flags[16 G + 16]; error word at [16 G + 32];
                       // class release words at flags[16 G + 64 + 16 c], c = 0..7
  unsigned long long* ts;   // FF_CHAIN_TRACE: [G][64][4] wall-clock stamps (op entered, work done, boundary passed) or null
};

__device__ __forceinline__ unsigned chain_ld(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_st(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// bounded poll: true when *p >= target was observed
__device__ __forceinline__ bool chain_wait_ge(const unsigned* p, unsigned target) {
  for (unsigned n = 0; n < CHAIN_SPIN_LIMIT; ++n) {
    if (chain_ld(p) >= target) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// grid-wide phase boundary (see the file header); epoch = 1, 2, 3, ... over the life of the sync words.
// Release side in two levels: workgroup 0 publishes the release word, the FIRST workgroup of each XCD residue class
// (blockIdx < 8) polls it and republishes into its class's word, which the other workgroups of the class poll.  A class =
// blockIdx & 7 = the XCD the dispatcher puts the workgroup on (placement only matters for speed: the class word then lives in
// that XCD's L2 and ~31 pollers are served there).  With every idle workgroup polling ONE word, 255 pollers saturate its
// memory channel (~90 accesses/us) for the whole phase: measured 8 us per 32x32x512 tile and 36 us per attention unit next
// to them, against 4-5 us alone.
__device__ __forceinline__ void chain_boundary(const ChainSync& s, int G, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave: its write-through stores have left the CU
  __syncthreads();
  unsigned* release = s.flags + 16 * G + 16;
  unsigned* err = s.flags + 16 * G + 32;
  unsigned* cls = s.flags + 16 * G + 64 + 16 * (blockIdx.x & 7);
  if (threadIdx.x == 0) chain_st(s.flags + 16 * blockIdx.x, epoch);
  if (blockIdx.x == 0) {
    bool ok = true;
    if (chain_ld(release) != CHAIN_POISON) {
      for (int b = threadIdx.x; b < G; b += blockDim.x) ok = chain_wait_ge(s.flags + 16 * b, epoch) && ok;
    }
    const int all_ok = __syncthreads_and(ok ? 1 : 0);
    if (threadIdx.x == 0) {
      if (!all_ok) { chain_st(err, 1u); chain_st(release, CHAIN_POISON); }
      else if (chain_ld(release) != CHAIN_POISON) chain_st(release, epoch);
    }
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x < 8) {
      if (!chain_wait_ge(release, epoch)) { chain_st(err, 2u); chain_st(release, CHAIN_POISON); }
      const unsigned r = chain_ld(release);
      chain_st(cls, r == CHAIN_POISON ? CHAIN_POISON : epoch);
    } else {
      if (!chain_wait_ge(cls, epoch)) { chain_st(err, 3u); chain_st(release, CHAIN_POISON); chain_st(cls, CHAIN_POISON); }
    }
  }
  __syncthreads();
}

// Every operator is its own (non-inlined) function: the register allocation of the attention unit (218 VGPRs alone) then does
// not have to carry the interpreter's state and the other operators' descriptors across its body (inlined, the launch
// spilled 65-96 VGPRs inside the operator bodies; as functions the only scratch traffic is the attention operator's
// callee-saved registers, once per operator).  The dynamic LDS block is re-declared inside each one, so LDS accesses stay
// ds_* instructions.
// The descriptor pointer arrives in VGPRs (function-call ABI) although it is the same for every lane: readfirstlane makes
// that provable again, so the descriptor is fetched with scalar loads and every pointer in it stays wave-uniform (SGPR base +
// per-lane offset addressing instead of 64-bit per-lane pointers).
__device__ __forceinline__ const FF_GLOBAL ff_chain_op* chain_uniform(const ff_chain_op* p) {
  const ff_u64 v = (ff_u64)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const FF_GLOBAL ff_chain_op*)(((ff_u64)hi << 32) | lo);
}

// descriptor part -> registers, word by word through the global pointer (C++ has no copy constructor across address spaces)
template <typename T>
__device__ __forceinline__ T chain_fetch(const FF_GLOBAL void* p) {
  static_assert(sizeof(T) % 4 == 0, "descriptor parts are whole words");
  T out;
  unsigned* o = reinterpret_cast<unsigned*>(&out);
  const FF_GLOBAL unsigned* q = (const FF_GLOBAL unsigned*)p;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) o[i] = q[i];
  return out;
}

template <int MODE>
__device__ __attribute__((noinline)) void chain_op_gemm(const ff_chain_op* opp_v) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FF_GLOBAL ff_chain_op* opp = chain_uniform(opp_v);
  const GemmArgs g = chain_fetch<GemmArgs>(&opp->u.g);
  const int units = opp->units;
  const int G = gridDim.x;
  const int per = g.tiles_m * g.tiles_n;
  for (int u = blockIdx.x; u < units; u += G) {
    const int bz = u / per, tile = u - bz * per;
    switch (g.K) {
      case 128: ff_gemm_small_tile<16, MODE, CHAIN_WAVES, true>(g, tile, bz, lds); break;
      case 256: ff_gemm_small_tile<32, MODE, CHAIN_WAVES, true>(g, tile, bz, lds); break;
      case 512: ff_gemm_small_tile<64, MODE, CHAIN_WAVES, true>(g, tile, bz, lds); break;
      default: ff_gemm_small_tile<128, MODE, CHAIN_WAVES, true>(g, tile, bz, lds); break;   // 1024 (checked on the host)
    }
    __syncthreads();   // the partial-tile area is rewritten by the next tile
  }
}

__device__ __attribute__((noinline)) void chain_op_attention(const ff_chain_op* opp_v) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const FF_GLOBAL ff_chain_op* opp = chain_uniform(opp_v);
  const ff_attn_desc d = chain_fetch<ff_attn_desc>(&opp->u.a);
  const int units = opp->units, q_tiles = opp->aux0, ks = opp->aux1;
  const long total_units = opp->total_units;
  const int G = gridDim.x;
  for (int vb = blockIdx.x; vb < units; vb += G) {
    ff_attention_wave_block<CHAIN_WAVES, true>(d, q_tiles, ks, total_units, 0, 0, (long)vb, lds);
    __syncthreads();   // a wave's K patch doubles as its combine record: all reads done before the next unit writes
  }
}

__device__ __attribute__((noinline)) void chain_op_layernorm(const ff_chain_op* opp_v) {
  const FF_GLOBAL ff_chain_op* opp = chain_uniform(opp_v);
  const LnArgs a = chain_fetch<LnArgs>(&opp->u.ln);
  const int nv = opp->aux0;
  const int G = gridDim.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int row = blockIdx.x * CHAIN_WAVES + wave; row < a.rows; row += G * CHAIN_WAVES) {
    if (nv <= 1) ff_layernorm_row<1, true>(a, row, lane);
    else if (nv <= 2) ff_layernorm_row<2, true>(a, row, lane);
    else if (nv <= 4) ff_layernorm_row<4, true>(a, row, lane);
    else ff_layernorm_row<8, true>(a, row, lane);
  }
}

__device__ __attribute__((noinline)) void chain_op_pointer(const ff_chain_op* opp_v) {
  const FF_GLOBAL ff_chain_op* opp = chain_uniform(opp_v);
  const PointerArgs a = chain_fetch<PointerArgs>(&opp->u.p);
  const int G = gridDim.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int b = blockIdx.x * CHAIN_WAVES + wave; b < a.B; b += G * CHAIN_WAVES) ff_pointer_reduce_row<true>(a, b, lane);
}

__global__ __launch_bounds__(CHAIN_THREADS, 1) void chain_kernel(const ff_chain_op* __restrict__ ops, int nops, ChainSync sync,
                                                                 unsigned epoch0) {
  const int G = gridDim.x;
  unsigned epoch = epoch0;
  for (int i = 0; i < nops; ++i) {
    const ff_chain_op* op = ops + i;
    const int kind = op->kind;
    if (sync.ts && i < 64 && threadIdx.x == 0) sync.ts[((size_t)blockIdx.x * 64 + i) * 4 + 0] = wall_clock64();
    if (kind == FF_CH_GEMM) {
      const int mode = op->u.g.ln_in ? 1 : (op->u.g.ln_out ? 2 : 0);
      if (mode == 1) chain_op_gemm<1>(op);
      else if (mode == 2) chain_op_gemm<2>(op);
      else chain_op_gemm<0>(op);
    } else if (kind == FF_CH_ATTN) {
      chain_op_attention(op);
    } else if (kind == FF_CH_LN) {
      chain_op_layernorm(op);
    } else if (kind == FF_CH_PTR) {
      chain_op_pointer(op);
    }
    if (sync.ts && i < 64 && threadIdx.x == 0) sync.ts[((size_t)blockIdx.x * 64 + i) * 4 + 1] = wall_clock64();
    if (op->barrier && i + 1 < nops) {
      ++epoch;
      chain_boundary(sync, G, epoch);
    }
    if (sync.ts && i < 64 && threadIdx.x == 0) sync.ts[((size_t)blockIdx.x * 64 + i) * 4 + 2] = wall_clock64();
  }
}

// ---- host side: per-device context and the recorder ---------------------------------------------------------------------
struct ChainCtx {
  bool ready = false;
  int G = 0;
  ff_chain_op* host_ops = nullptr;   // pinned
  ff_chain_op* dev_ops = nullptr;
  size_t cap_ops = 0, used_ops = 0;
  unsigned* flags = nullptr;
  unsigned epoch = 0;
  bool attr_done = false;
  unsigned long long* ts = nullptr;   // FF_CHAIN_TRACE
  int traced = 0;
  int flow_launches = 0;
  unsigned* flow_ctr = nullptr;       // flow launches: [2][FF_FLOW_MAX_OPS][FLOW_PANELS] arrival / consumer counters + error word
};
constexpr int FLOW_PANELS = 1 << 16;   // row panels of 64 rows a flow launch can track (4 M rows)
constexpr size_t FLOW_WORDS = (size_t)2 * FF_FLOW_MAX_OPS * FLOW_PANELS + 64;
constexpr int CHAIN_MAX_DEV = 16;
ChainCtx g_ctx[CHAIN_MAX_DEV];
std::mutex g_chain_mu;

struct Recorder {
  bool active = false;
  bool flow = false;             // flow recording: only 64x64-tile projections, each depending on the one before
  bool failed = false;
  bool skip_barrier = false;     // hint: the NEXT recorded operator does not depend on the previous one
  double flops = 0;
  std::vector<ff_chain_op> ops;
};
thread_local Recorder t_rec;

int ctx_get(ChainCtx** out) {
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  FF_CHECK_ARG(dev >= 0 && dev < CHAIN_MAX_DEV, "ff_chain: device index %d out of range", dev);
  *out = &g_ctx[dev];
  return FF_OK;
}

}  // namespace

static int flow_max_ops() {   // FF_FLOW_MAX_OPS=<n>: debugging aid (operators per flow launch)
  static const int v = getenv("FF_FLOW_MAX_OPS") ? atoi(getenv("FF_FLOW_MAX_OPS")) : FF_FLOW_MAX_OPS;
  return v < 1 ? 1 : (v > FF_FLOW_MAX_OPS ? FF_FLOW_MAX_OPS : v);
}
bool ff_chain_recording() { return t_rec.active; }
bool ff_flow_recording() { return t_rec.active && t_rec.flow; }
void ff_chain_next_is_independent() { if (t_rec.active) t_rec.skip_barrier = true; }

static void chain_push(ff_chain_op& op, double flops) {
  op.barrier = 1;
  if (t_rec.skip_barrier && !t_rec.ops.empty()) t_rec.ops.back().barrier = 0;
  t_rec.skip_barrier = false;
  t_rec.flops += flops;
  t_rec.ops.push_back(op);
}

bool ff_chain_gemm_ok(const GemmArgs& g, int batch) {
  const bool k_ok = g.K == 128 || g.K == 256 || g.K == 512 || g.K == 1024;
  const long tiles = (long)ff_cdiv(g.M, 32) * ff_cdiv(g.N, 32) * batch;
  return k_ok && (!g.A2 || (g.n_split % 32) == 0) && tiles < (1L << 30) && !(g.ln_in && g.ln_out) &&
         (!g.ln_in || (g.ln_nseg >= 2 && g.ln_nseg <= 16 && (g.ln_nseg & 1) == 0));
}

int ff_chain_record_gemm(const GemmArgs& g_in, int batch) {
  if (t_rec.flow) {   // one 64x64-tile operator of a flow launch
    const bool ok = batch == 1 && (g_in.K % 64) == 0 && g_in.K >= 128 && !g_in.A2 && !(g_in.ln_in && g_in.ln_out) &&
                    (!g_in.ln_out || (g_in.N & 31) == 0) && (!g_in.ln_in || g_in.ln_nseg <= 16) &&
                    (int)t_rec.ops.size() < flow_max_ops() && ff_cdiv(g_in.M, 64) <= FLOW_PANELS &&
                    (t_rec.ops.empty() || t_rec.ops.back().u.g.M == g_in.M);
    if (!ok) { t_rec.failed = true; return FF_OK; }
    ff_chain_op op;
    memset(&op, 0, sizeof(op));
    op.kind = FF_CH_GEMM64;
    op.u.g = g_in;
    op.u.g.tiles_m = ff_cdiv(g_in.M, 64);
    op.u.g.tiles_n = ff_cdiv(g_in.N, 64);
    op.units = op.u.g.tiles_m * op.u.g.tiles_n;
    chain_push(op, 2.0 * g_in.M * g_in.N * g_in.K);
    return FF_OK;
  }
  if (!ff_chain_gemm_ok(g_in, batch)) { t_rec.failed = true; return FF_OK; }
  ff_chain_op op;
  memset(&op, 0, sizeof(op));
  op.kind = FF_CH_GEMM;
  op.u.g = g_in;
  op.u.g.tiles_m = ff_cdiv(g_in.M, 32);
  op.u.g.tiles_n = ff_cdiv(g_in.N, 32);
  op.units = op.u.g.tiles_m * op.u.g.tiles_n * batch;
  chain_push(op, 2.0 * g_in.M * g_in.N * g_in.K * batch);
  return FF_OK;
}

int ff_chain_record_attention(const ff_attn_desc& d) {
  if (t_rec.flow) { t_rec.failed = true; return FF_OK; }
  ff_chain_op op;
  memset(&op, 0, sizeof(op));
  op.kind = FF_CH_ATTN;
  op.u.a = d;
  const long gh = (long)d.num_groups * d.num_heads;
  const int qt = ff_cdiv(d.nq, 32);
  const long units = gh * qt;
  const int key_tiles = ff_cdiv(d.nk, 32);
  // key tiles of a unit are dealt to ks waves while the launch would otherwise leave wave slots idle
  int ks = 1;
  while (ks < CHAIN_WAVES && units * ks * 2 <= 2048 && ks < key_tiles) ks *= 2;
  const long per_block = CHAIN_WAVES / ks;
  const long nblocks = (units + per_block - 1) / per_block;
  if (nblocks >= (1L << 30)) { t_rec.failed = true; return FF_OK; }
  op.aux0 = qt; op.aux1 = ks;
  op.total_units = units;
  op.units = (int)nblocks;
  chain_push(op, 4.0 * FF_HEAD_DIM * (double)gh * d.nq * d.nk);
  return FF_OK;
}

int ff_chain_record_layernorm(const LnArgs& a) {
  if (t_rec.flow) { t_rec.failed = true; return FF_OK; }
  ff_chain_op op;
  memset(&op, 0, sizeof(op));
  op.kind = FF_CH_LN;
  op.u.ln = a;
  op.aux0 = ff_cdiv(a.E / 4, 64);
  op.units = a.rows;
  chain_push(op, 0.0);
  return FF_OK;
}

int ff_chain_record_pointer(const PointerArgs& a) {
  if (t_rec.flow) { t_rec.failed = true; return FF_OK; }
  ff_chain_op op;
  memset(&op, 0, sizeof(op));
  op.kind = FF_CH_PTR;
  op.u.p = a;
  op.units = a.B;
  chain_push(op, 0.0);
  return FF_OK;
}

// Make room for `ops_needed` descriptors in the context of the current device and reset its ring and sync words (called once
// per decode, before anything is enqueued; the previous decode on this device has been synchronised).
int ff_chain_prepare(size_t ops_needed, hipStream_t st) {
  ChainCtx* c;
  FF_RETURN_IF(ctx_get(&c));
  std::lock_guard<std::mutex> lock(g_chain_mu);
  if (!c->ready) {
    int dev = 0, cus = 0;
    FF_CHECK_HIP(hipGetDevice(&dev));
    FF_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    c->G = cus < 256 ? (cus > 0 ? cus : 1) : 256;
    FF_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&c->flags), (size_t)(c->G + 16) * 64));
    FF_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&c->flow_ctr), FLOW_WORDS * sizeof(unsigned)));
    c->ready = true;
  }
  if (ops_needed > c->cap_ops) {
    FF_CHECK_HIP(hipDeviceSynchronize());
    if (c->host_ops) (void)hipHostFree(c->host_ops);
    if (c->dev_ops) (void)hipFree(c->dev_ops);
    c->host_ops = nullptr; c->dev_ops = nullptr; c->cap_ops = 0;
    FF_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->host_ops), ops_needed * sizeof(ff_chain_op), hipHostMallocDefault));
    FF_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&c->dev_ops), ops_needed * sizeof(ff_chain_op)));
    c->cap_ops = ops_needed;
  }
  c->used_ops = 0;
  c->epoch = 0;
  FF_CHECK_HIP(hipMemsetAsync(c->flags, 0, (size_t)(c->G + 16) * 64, st));
  FF_CHECK_HIP(hipMemsetAsync(c->flow_ctr, 0, FLOW_WORDS * sizeof(unsigned), st));
  return FF_OK;
}

int ff_flow_begin() {
  FF_RETURN_IF(ff_chain_begin());
  t_rec.flow = true;
  return FF_OK;
}

// Ends a flow recording and enqueues ONE gemm_flow_kernel launch (ff_gemm.hip); *launched = 0: nothing enqueued.
int ff_flow_end(hipStream_t st, int* launched) {
  *launched = 0;
  t_rec.active = false;
  t_rec.flow = false;
  if (t_rec.failed || t_rec.ops.empty()) { t_rec.ops.clear(); return FF_OK; }
  ChainCtx* c;
  FF_RETURN_IF(ctx_get(&c));
  const size_t n = t_rec.ops.size();
  if (!c->ready || c->used_ops + n > c->cap_ops) { t_rec.ops.clear(); return FF_OK; }
  ff_chain_op* h = c->host_ops + c->used_ops;
  ff_chain_op* d = c->dev_ops + c->used_ops;
  memcpy(h, t_rec.ops.data(), n * sizeof(ff_chain_op));
  c->used_ops += n;
  FF_CHECK_HIP(hipMemcpyAsync(d, h, n * sizeof(ff_chain_op), hipMemcpyHostToDevice, st));
  unsigned* ctr = c->flow_ctr;
  unsigned* done = c->flow_ctr + (size_t)FF_FLOW_MAX_OPS * FLOW_PANELS;
  unsigned* err = c->flow_ctr + (size_t)2 * FF_FLOW_MAX_OPS * FLOW_PANELS;
  {
    FFProfScope prof(FF_CAT_GEMM, t_rec.flops, st);
    double bytes = 0;
    for (const ff_chain_op& op : t_rec.ops) {
      const GemmArgs& g = op.u.g;
      bytes += 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N * (g.res ? 2 : 1));
    }
    ff_prof_add_bytes(FF_CAT_GEMM, bytes);
    FF_RETURN_IF(ff_gemm_flow_launch(d, (int)n, ctr, done, err, FLOW_PANELS, st));
  }
  c->flow_launches++;
  t_rec.ops.clear();
  *launched = 1;
  return FF_OK;
}

int ff_chain_begin() {
  t_rec.flow = false;
  t_rec.active = true;
  t_rec.failed = false;
  t_rec.skip_barrier = false;
  t_rec.flops = 0;
  t_rec.ops.clear();
  return FF_OK;
}

void ff_chain_abort() {
  t_rec.active = false;
  t_rec.flow = false;
  t_rec.ops.clear();
}

// Ends the recording and enqueues ONE launch that runs the recorded operators.  *launched = 0 when nothing was launched
// (an operator did not fit the chain forms, or the ring is full): the caller then enqueues the operators one by one.
int ff_chain_end(hipStream_t st, int* launched) {
  *launched = 0;
  t_rec.active = false;
  if (t_rec.failed || t_rec.ops.empty()) { t_rec.ops.clear(); return FF_OK; }
  ChainCtx* c;
  FF_RETURN_IF(ctx_get(&c));
  const size_t n = t_rec.ops.size();
  if (!c->ready || c->used_ops + n > c->cap_ops) { t_rec.ops.clear(); return FF_OK; }
  if (!c->attr_done) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     chain_lds_floats() * (int)sizeof(float)));
    c->attr_done = true;
  }
  ff_chain_op* h = c->host_ops + c->used_ops;
  ff_chain_op* d = c->dev_ops + c->used_ops;
  memcpy(h, t_rec.ops.data(), n * sizeof(ff_chain_op));
  c->used_ops += n;
  FF_CHECK_HIP(hipMemcpyAsync(d, h, n * sizeof(ff_chain_op), hipMemcpyHostToDevice, st));
  unsigned barriers = 0;
  for (size_t i = 0; i + 1 < n; ++i) barriers += t_rec.ops[i].barrier ? 1u : 0u;
  // FF_CHAIN_TRACE=<k>: wall-clock stamps of the k-th chain launch of the process (per workgroup and operator), printed to stderr
  static const int trace_at = getenv("FF_CHAIN_TRACE") ? atoi(getenv("FF_CHAIN_TRACE")) : -1;
  const bool trace = trace_at >= 0 && c->traced++ == trace_at;
  if (trace && !c->ts) FF_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&c->ts), (size_t)c->G * 64 * 4 * sizeof(unsigned long long)));
  if (trace) FF_CHECK_HIP(hipMemsetAsync(c->ts, 0, (size_t)c->G * 64 * 4 * sizeof(unsigned long long), st));
  ChainSync s{c->flags, trace ? c->ts : nullptr};
  {
    FFProfScope prof(FF_CAT_CHAIN, t_rec.flops, st);
    hipLaunchKernelGGL(chain_kernel, dim3(c->G), dim3(CHAIN_THREADS), chain_lds_floats() * sizeof(float), st, d, (int)n, s,
                       c->epoch);
    FF_CHECK_LAUNCH();
  }
  if (trace) {
    std::vector<unsigned long long> h((size_t)c->G * 64 * 4);
    FF_CHECK_HIP(hipStreamSynchronize(st));
    FF_CHECK_HIP(hipMemcpy(h.data(), c->ts, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long t0 = h[0];
    fprintf(stderr, "[chain trace] launch %d: %zu operators, G = %d (times in us since workgroup 0 entered operator 0; 100 MHz clock)\n",
            trace_at, n, c->G);
    for (size_t i = 0; i < n && i < 64; ++i) {
      double work_max = 0, work_min = 1e30, enter_max = 0, leave_max = 0; int busy = 0;
      for (int b = 0; b < c->G; ++b) {
        const unsigned long long* r = &h[((size_t)b * 64 + i) * 4];
        const double w = (double)(r[1] - r[0]) * 0.01;
        work_max = w > work_max ? w : work_max; work_min = w < work_min ? w : work_min;
        const double e = (double)(r[0] - t0) * 0.01, l = (double)(r[2] - t0) * 0.01;
        enter_max = e > enter_max ? e : enter_max; leave_max = l > leave_max ? l : leave_max;
      }
      const ff_chain_op& op = t_rec.ops[i];
      busy = op.units < c->G ? op.units : c->G;
      const unsigned long long* r0 = &h[i * 4];
      fprintf(stderr, "  op %2zu kind %d units %5d (%3d busy wgs) barrier %d: wg0 enter %8.2f work %6.2f boundary %6.2f | all wgs: work min %6.2f max %6.2f, "
                      "last enter %8.2f last leave %8.2f\n", i, op.kind, op.units, busy, op.barrier, (double)(r0[0] - t0) * 0.01,
              (double)(r0[1] - r0[0]) * 0.01, (double)(r0[2] - r0[1]) * 0.01, work_min, work_max, enter_max, leave_max);
    }
  }
  c->epoch += barriers;
  t_rec.ops.clear();
  *launched = 1;
  return FF_OK;
}

// error word of the current device's sync area (0 = no boundary ever timed out); synchronises `st`
int ff_chain_check(hipStream_t st) {
  ChainCtx* c;
  FF_RETURN_IF(ctx_get(&c));
  if (!c->ready) return FF_OK;
  unsigned err = 0, ferr = 0;
  FF_CHECK_HIP(hipMemcpyAsync(&err, c->flags + 16 * c->G + 32, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  FF_CHECK_HIP(hipMemcpyAsync(&ferr, c->flow_ctr + (size_t)2 * FF_FLOW_MAX_OPS * FLOW_PANELS, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  FF_CHECK_HIP(hipStreamSynchronize(st));
  if (ferr != 0) {
    ff_set_error("ff_decode: a row-panel dependency of a flow launch timed out (code %u): its workgroups were not all resident "
                 "(another process on the device?); re-run without FF_FLOW", ferr);
    return FF_ERR_LAUNCH;
  }
  if (err != 0) {
    ff_set_error("ff_decode: a grid-wide phase boundary of the chain kernel timed out (code %u): the workgroups of the launch were "
                 "not all resident (another process on the device?); re-run without FF_CHAIN", err);
    return FF_ERR_LAUNCH;
  }
  return FF_OK;
}

#endif  // FF_EXPERIMENTAL
