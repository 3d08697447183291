// Row-wise kernels of the decode path: LayerNorm(+pos), memory+pos, feedback gather, embedding
// assembly.  All are HBM/L2-bandwidth bound: one wavefront per row, 16-byte loads/stores,
// butterfly reductions through __shfl_xor (no LDS).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "ff_common.h"
#include "ff_device.h"

// ---- library-level helpers ---------------------------------------------------------------------
static thread_local char g_ff_error[512] = "";

void ff_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_ff_error, sizeof(g_ff_error), fmt, ap);
  va_end(ap);
}

int ff_num_cus() {
  static std::atomic<int> cached[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  int n = cached[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// ---- tuning knobs -------------------------------------------------------------------------------------------------------------------
namespace {
struct KnobDef { const char* name; int def; bool presence; };   // presence: the variable being set at all means 1
const KnobDef g_knob_defs[FF_K_COUNT] = {
    {"FF_L0_FOLD", 1, false}, {"FF_POINTER_FOLD", 1, false}, {"FF_LAST_QKV_ONE_LAUNCH_ROWS", 512, false},
    {"FF_PINNED_COUNTERS", 65536, false}, {"FF_DEBUG_TIMING", 0, true},
    {"FF_DMA_MIN_ROWS", 4096, false}, {"FF_DMA_MIN_ROWS_N512", 7680, false}, {"FF_DMA_MIN_ROWS_WIDE", 2560, false},
    {"FF_SK_HYBRID", 1, false}, {"FF_SK_HYBRID_FIX", 10, false}, {"FF_SK_HYBRID_MAXLEFT8", 4, false},
    {"FF_SK_HYBRID_MINU", 2, false}, {"FF_SK_HYBRID_FORCE", 0, false},
    {"FF_NO_PANEL", 0, true}, {"FF_X3_SMALL_SPLIT", 0, false},
    {"FF_RK_SPLIT_OLD", 1, false}, {"FF_RK_SPLIT_YOUNG", 1, false}, {"FF_RK_PHASE", 0, false}, {"FF_RK_ROTATE", 1, false},
    {"FF_X3_NEED_N1024", 7, false}, {"FF_X3_NEED_N512", 11, false}, {"FF_X2H_ATTN", 1, false},
};
std::atomic<int> g_knobs[FF_K_COUNT];
std::once_flag g_knobs_once;
void knobs_init() {
  std::call_once(g_knobs_once, [] {
    for (int i = 0; i < FF_K_COUNT; ++i) {
      const char* e = getenv(g_knob_defs[i].name);
      g_knobs[i].store(e ? (g_knob_defs[i].presence ? 1 : atoi(e)) : g_knob_defs[i].def, std::memory_order_relaxed);
    }
  });
}
int knob_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < FF_K_COUNT; ++i)
    if (strcmp(name, g_knob_defs[i].name) == 0) return i;
  return -1;
}
}  // namespace

int ff_knob(int id) {
  knobs_init();
  return g_knobs[id].load(std::memory_order_relaxed);
}

extern "C" int ff_set_tuning(const char* name, int value) {
  const int i = knob_index(name);
  FF_CHECK_ARG(i >= 0, "ff_set_tuning: unknown knob '%s'", name ? name : "(null)");
  FF_CHECK_ARG(i != FF_K_PINNED_COUNTERS || (value > 0 && value <= 65536), "ff_set_tuning: FF_PINNED_COUNTERS must be in 1..65536");
  knobs_init();
  g_knobs[i].store(value, std::memory_order_relaxed);
  ff_tuning_changed();
  return FF_OK;
}

extern "C" int ff_get_tuning(const char* name, int* value) {
  const int i = knob_index(name);
  FF_CHECK_ARG(i >= 0 && value, "ff_get_tuning: unknown knob '%s'", name ? name : "(null)");
  *value = ff_knob(i);
  return FF_OK;
}

extern "C" int ff_reset_tuning(void) {   // back to the defaults (NOT the environment's values): tests
  knobs_init();
  for (int i = 0; i < FF_K_COUNT; ++i) g_knobs[i].store(g_knob_defs[i].def, std::memory_order_relaxed);
  ff_tuning_changed();
  return FF_OK;
}

extern "C" int ff_version(void) { return FF_ABI_VERSION; }
extern "C" const char* ff_last_error(void) { return g_ff_error; }
extern "C" int ff_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

// ---- event profiler ------------------------------------------------------------------------------
#include <vector>
namespace {
struct ProfRec { int cat; double work; hipEvent_t a, b; };
struct Profiler {
  bool enabled = false;
  std::vector<hipEvent_t> pool;
  size_t next = 0;
  std::vector<ProfRec> recs;
  double bytes[FF_NUM_CAT] = {};  // algorithmic operand + result bytes per category
  hipEvent_t get() {
    if (next == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[next++];
  }
} g_prof;
}  // namespace

bool ff_prof_enabled() { return g_prof.enabled; }

static std::atomic<unsigned long long> g_tuning_epoch{0};
unsigned long long ff_tuning_epoch() { return g_tuning_epoch.load(); }
void ff_tuning_changed() { g_tuning_epoch.fetch_add(1); }
void ff_prof_open(int cat, double work, hipStream_t st) {
  ProfRec r{cat, work, g_prof.get(), g_prof.get()};
  if (r.a) (void)hipEventRecord(r.a, st);
  g_prof.recs.push_back(r);
}
void ff_prof_add_bytes(int cat, double bytes) {
  if (g_prof.enabled && cat >= 0 && cat < FF_NUM_CAT) g_prof.bytes[cat] += bytes;
}
void ff_prof_close(hipStream_t st) {
  if (!g_prof.recs.empty() && g_prof.recs.back().b) (void)hipEventRecord(g_prof.recs.back().b, st);
}

extern "C" int ff_profile_begin(void) {
  g_prof.recs.clear();
  g_prof.next = 0;
  for (double& b : g_prof.bytes) b = 0;
  g_prof.enabled = true;
  return FF_OK;
}

extern "C" int ff_profile_bytes(double* bytes_by_cat, int ncat) {
  for (int c = 0; c < ncat; ++c) bytes_by_cat[c] = c < FF_NUM_CAT ? g_prof.bytes[c] : 0.0;
  return FF_OK;
}

extern "C" int ff_profile_end(double* ms, double* work, long long* launches, int ncat) {
  g_prof.enabled = false;
  FF_CHECK_HIP(hipDeviceSynchronize());
  for (int c = 0; c < ncat; ++c) { ms[c] = 0; work[c] = 0; launches[c] = 0; }
  for (const ProfRec& r : g_prof.recs) {
    if (r.cat < 0 || r.cat >= ncat || !r.a || !r.b) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { (void)hipGetLastError(); continue; }
    ms[r.cat] += t;
    work[r.cat] += r.work;
    launches[r.cat] += 1;
  }
  g_prof.recs.clear();
  g_prof.next = 0;
  return FF_OK;
}

// What the event bracket adds to a measured interval: `launches` empty kernels are queued back to back, each between
// its own event pair exactly like ff_prof_open / ff_prof_close do; the mean (b - a) interval of an EMPTY kernel is what
// a category time of ff_profile_end carries per launch on top of the kernel's own duration.
__global__ void ff_empty_kernel() {}
extern "C" int ff_profile_bracket_us(int launches, double* us_per_launch, ff_stream_t stream) {
  FF_CHECK_ARG(launches > 0 && launches <= 4096 && us_per_launch, "ff_profile_bracket_us: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(2 * (size_t)launches);
  for (hipEvent_t& e : ev) FF_CHECK_HIP(hipEventCreate(&e));
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(ff_empty_kernel, dim3(1), dim3(64), 0, st);   // warm
  for (int i = 0; i < launches; ++i) {
    FF_CHECK_HIP(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(ff_empty_kernel, dim3(1), dim3(64), 0, st);
    FF_CHECK_HIP(hipEventRecord(ev[2 * i + 1], st));
  }
  FF_CHECK_HIP(hipStreamSynchronize(st));
  double tot = 0;
  for (int i = 0; i < launches; ++i) {
    float t = 0.f;
    FF_CHECK_HIP(hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]));
    tot += t;
  }
  for (hipEvent_t& e : ev) (void)hipEventDestroy(e);
  *us_per_launch = 1e3 * tot / launches;
  return FF_OK;
}

// ---- effective shader clock --------------------------------------------------------------------------------------------
// One lane reads the shader-clock counter (s_memtime: ticks at the clock the SIMDs run at, MI355X_MICROARCH.md) and the
// constant 100 MHz counter (s_memrealtime) when it starts, sleeps until `spin_ticks` of the constant counter have passed,
// and reads both again: shader cycles / wall time = the clock the chip ran at in between.  Launched on a stream of its own
// BESIDE the kernels under test (a 64-thread block fits next to their workgroups), it samples the clock UNDER that load.
__global__ void clock_probe_kernel(unsigned long long* out, unsigned long long spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long w0 = wall_clock64();
  const unsigned long long c0 = clock64();
  unsigned long long w1 = w0;
  while (w1 - w0 < spin_ticks) {
    __builtin_amdgcn_s_sleep(64);
    w1 = wall_clock64();
  }
  const unsigned long long c1 = clock64();
  out[0] = c1 - c0;
  out[1] = w1 - w0;
}
namespace {
unsigned long long* g_clock_out = nullptr;   // host-mapped pinned [2]
std::mutex g_clock_mu;
}  // namespace
extern "C" int ff_clock_probe_launch(double spin_us, ff_stream_t stream) {
  FF_CHECK_ARG(spin_us > 0 && spin_us <= 5e6, "ff_clock_probe_launch: spin_us in (0, 5e6]");
  std::lock_guard<std::mutex> lock(g_clock_mu);
  if (!g_clock_out)
    FF_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&g_clock_out), 2 * sizeof(unsigned long long),
                               hipHostMallocMapped | hipHostMallocCoherent));
  g_clock_out[0] = g_clock_out[1] = 0;
  unsigned long long* dev_out = nullptr;
  FF_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev_out), g_clock_out, 0));
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dev_out,
                     (unsigned long long)(spin_us * 100.0));   // 100 MHz constant counter
  FF_CHECK_LAUNCH();
  return FF_OK;
}
extern "C" int ff_clock_probe_read(double* ghz, double* measured_us, ff_stream_t stream) {
  FF_CHECK_ARG(ghz != nullptr, "ff_clock_probe_read: null output");
  FF_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::lock_guard<std::mutex> lock(g_clock_mu);
  FF_CHECK_ARG(g_clock_out && g_clock_out[1] > 0, "ff_clock_probe_read: no finished probe");
  const double us = (double)g_clock_out[1] / 100.0;
  *ghz = (double)g_clock_out[0] / us * 1e-3;
  if (measured_us) *measured_us = us;
  return FF_OK;
}

// ---- LayerNorm (+pos) --------------------------------------------------------------------------
// NV = float4 chunks per lane (E <= 256*NV).  Two-pass statistics in registers (mean, then the
// centred second moment) -- the same formula torch's CPU kernel evaluates, biased variance.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  ff_layernorm_row<NV>(a, row, lane);
}

extern "C" int ff_layernorm(const float* x, int ldx, const float* gamma, const float* beta,
                            float eps, float* y, int ldy, float* ypos, int ldypos,
                            const float* pos, int ldpos, int pos_div, int pos_mod, int rows, int E,
                            ff_stream_t stream) {
  if (rows == 0) return FF_OK;
  FF_CHECK_ARG(rows > 0 && E > 0 && (E & 3) == 0 && E <= 2048, "ff_layernorm: bad rows=%d E=%d", rows, E);
  FF_CHECK_ARG(x && gamma && beta && (y || ypos), "ff_layernorm: null pointer");
  FF_CHECK_ARG((ldx & 3) == 0 && ff_aligned16(x) && ff_aligned16(gamma) && ff_aligned16(beta),
               "ff_layernorm: x/gamma/beta must be 16-byte aligned, ld %% 4 == 0");
  FF_CHECK_ARG(!y || ((ldy & 3) == 0 && ff_aligned16(y)), "ff_layernorm: y misaligned");
  if (ypos) {
    FF_CHECK_ARG(pos && pos_div > 0 && pos_mod > 0 && (ldpos & 3) == 0 && (ldypos & 3) == 0 &&
                     ff_aligned16(pos) && ff_aligned16(ypos),
                 "ff_layernorm: bad pos arguments");
  }
  hipStream_t st = (hipStream_t)stream;
  const LnArgs la{x, ldx, gamma, beta, eps, y, ldy, ypos, ldypos, pos, ldpos, pos_div, pos_mod, rows, E};
  FFProfScope prof(FF_CAT_LN, (double)rows * E * 4.0 * (1 + (y != nullptr) + (ypos != nullptr)), st);
  dim3 block(256), grid(ff_cdiv(rows, 4));
  const int nv = ff_cdiv(E / 4, 64);
#define FF_LN_LAUNCH(NV) hipLaunchKernelGGL(layernorm_kernel<NV>, grid, block, 0, st, la)
  if (nv <= 1) FF_LN_LAUNCH(1);
  else if (nv <= 2) FF_LN_LAUNCH(2);
  else if (nv <= 4) FF_LN_LAUNCH(4);
  else FF_LN_LAUNCH(8);
#undef FF_LN_LAUNCH
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- out = x + pos[(row / div) % mod] ----------------------------------------------------------
__global__ __launch_bounds__(256) void add_pos_kernel(const float* __restrict__ x, int ldx,
                                                      const float* __restrict__ pos, int ldpos,
                                                      int pos_div, int pos_mod,
                                                      float* __restrict__ out, int ldout, int rows,
                                                      int nvec) {
  const size_t total = (size_t)rows * nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int row = (int)(i / nvec), vi = (int)(i % nvec);
    f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + vi * 4);
    f32x4 p = *reinterpret_cast<const f32x4*>(pos + (size_t)((row / pos_div) % pos_mod) * ldpos + vi * 4);
    *reinterpret_cast<f32x4*>(out + (size_t)row * ldout + vi * 4) = a + p;
  }
}

extern "C" int ff_add_pos(const float* x, int ldx, const float* pos, int ldpos, int pos_div,
                          int pos_mod, float* out, int ldout, int rows, int E, ff_stream_t stream) {
  if (rows == 0) return FF_OK;
  FF_CHECK_ARG(rows > 0 && E > 0 && (E & 3) == 0 && pos_div > 0 && pos_mod > 0, "ff_add_pos: bad sizes");
  FF_CHECK_ARG(x && pos && out && ((ldx | ldpos | ldout) & 3) == 0 && ff_aligned16(x) &&
                   ff_aligned16(pos) && ff_aligned16(out), "ff_add_pos: bad pointers");
  size_t total = (size_t)rows * (E / 4);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  FFProfScope prof(FF_CAT_ROWOP, (double)rows * E * 8.0, (hipStream_t)stream);
  hipLaunchKernelGGL(add_pos_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, pos,
                     ldpos, pos_div, pos_mod, out, ldout, rows, E / 4);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- x = gelu(x), exact erf form (torch F.gelu default; reference transformer.py:276-284 "gelu") ------------------------
// Module surface only: no reference config uses it, so it is a row op behind the plain projection and not a fourth epilogue
// form inside the hand-scheduled projection kernels.
__global__ __launch_bounds__(256) void gelu_kernel(float* __restrict__ x, int ldx, int rows, int E4) {
  const size_t total = (size_t)rows * E4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / E4), c = (int)(i % E4);
    f32x4* p = reinterpret_cast<f32x4*>(x + (size_t)r * ldx) + c;
    f32x4 v = *p;
    v.x = 0.5f * v.x * (1.f + erff(v.x * 0.70710678118654752440f));
    v.y = 0.5f * v.y * (1.f + erff(v.y * 0.70710678118654752440f));
    v.z = 0.5f * v.z * (1.f + erff(v.z * 0.70710678118654752440f));
    v.w = 0.5f * v.w * (1.f + erff(v.w * 0.70710678118654752440f));
    *p = v;
  }
}
// out[c, r] = in[r, c]: 32 x 32 tiles through LDS (padded), coalesced both ways
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, int ld_in, int rows, int cols,
                                                        float* __restrict__ out, int ld_out) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(size_t)(r0 + i) * ld_in + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * ld_out + r0 + tx] = tile[tx][i];
}

int ff_transpose(const float* in, int ld_in, int rows, int cols, float* out, int ld_out, hipStream_t st) {
  FF_CHECK_ARG(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, "ff_transpose: bad arguments");
  FFProfScope prof(FF_CAT_ROWOP, (double)rows * cols * 8.0, st);
  hipLaunchKernelGGL(transpose_kernel, dim3(ff_cdiv(cols, 32), ff_cdiv(rows, 32)), dim3(256), 0, st, in, ld_in, rows, cols, out, ld_out);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

extern "C" int ff_gelu(float* x, int ldx, int rows, int E, ff_stream_t stream) {
  if (rows == 0) return FF_OK;
  FF_CHECK_ARG(rows > 0 && E > 0 && (E & 3) == 0 && x && (ldx & 3) == 0 && ldx >= E && ff_aligned16(x), "ff_gelu: bad arguments");
  const size_t total = (size_t)rows * (E / 4);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  FFProfScope prof(FF_CAT_ROWOP, (double)rows * E * 8.0, (hipStream_t)stream);
  hipLaunchKernelGGL(gelu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, E / 4);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- out[b,:] = memory[b / spg, tok[b], :] -----------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ memory, int S,
                                                          int E, const int* __restrict__ tok, int B,
                                                          int spg, float* __restrict__ out,
                                                          int ldout) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  int t = tok[b];
  t = t < 0 ? 0 : (t >= S ? S - 1 : t);  // the reference would raise from torch.gather; clamp here
  const float* src = memory + ((size_t)(b / spg) * S + t) * E;
  float* dst = out + (size_t)b * ldout;
  for (int vi = lane; vi < (E >> 2); vi += 64)
    *reinterpret_cast<f32x4*>(dst + vi * 4) = *reinterpret_cast<const f32x4*>(src + vi * 4);
}

extern "C" int ff_gather_rows(const float* memory, int S, int E, const int* tok, int B,
                              int seqs_per_group, float* out, int ldout, ff_stream_t stream) {
  if (B == 0) return FF_OK;
  FF_CHECK_ARG(B > 0 && S > 0 && E > 0 && (E & 3) == 0 && seqs_per_group > 0 && (ldout & 3) == 0,
               "ff_gather_rows: bad sizes");
  FF_CHECK_ARG(memory && tok && out && ff_aligned16(memory) && ff_aligned16(out), "ff_gather_rows: bad pointers");
  FFProfScope prof(FF_CAT_ROWOP, (double)B * E * 8.0, (hipStream_t)stream);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ff_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream,
                     memory, S, E, tok, B, seqs_per_group, out, ldout);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- embedding assembly: token rows ++ edge rows ------------------------------------------------
__global__ __launch_bounds__(256) void assemble_embedding_kernel(
    const float* __restrict__ tok_embed, int num_token, const float* __restrict__ edge, int ld_edge,
    int N, int L, int E, float* __restrict__ out) {
  const int S = L + num_token;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N * S) return;
  const int n = row / S, s = row % S;
  const float* src = (s < num_token) ? tok_embed + (size_t)s * E
                                     : edge + ((size_t)n * L + (s - num_token)) * ld_edge;
  float* dst = out + (size_t)row * E;
  for (int vi = lane; vi < (E >> 2); vi += 64)
    *reinterpret_cast<f32x4*>(dst + vi * 4) = *reinterpret_cast<const f32x4*>(src + vi * 4);
}

extern "C" int ff_assemble_embedding(const float* tok_embed, int num_token, const float* edge_embed,
                                     int ld_edge, int N, int L, int E, float* out,
                                     ff_stream_t stream) {
  FF_CHECK_ARG(N > 0 && L >= 0 && num_token >= 0 && E > 0 && (E & 3) == 0 && (ld_edge & 3) == 0,
               "ff_assemble_embedding: bad sizes");
  FF_CHECK_ARG(tok_embed && out && (L == 0 || edge_embed) && ff_aligned16(tok_embed) &&
                   ff_aligned16(out) && ff_aligned16(edge_embed), "ff_assemble_embedding: bad pointers");
  int rows = N * (L + num_token);
  FFProfScope prof(FF_CAT_ROWOP, (double)rows * E * 8.0, (hipStream_t)stream);
  hipLaunchKernelGGL(assemble_embedding_kernel, dim3(ff_cdiv(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, tok_embed, num_token, edge_embed, ld_edge, N, L, E, out);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- process_masks + kv_len in one launch -----------------------------------------------------------------------------
// mask_out[n, 0:num_token] = 0, mask_out[n, num_token + l] = (in[n, l] != 0)   (reference model.py:61-69 / model_para.py:62-70:
// special tokens are never masked), kv_len[n] = 1 + the last unmasked key of row n (0 if every key is masked).  One wave per
// wireframe.  Replaces eight torch operators in front of ff_encode (zeros, type_as, cat, to(uint8), arange, ==, *, amax).
__global__ __launch_bounds__(256) void prepare_mask_kernel(const unsigned char* __restrict__ in, int N, int L, int num_token,
                                                           unsigned char* __restrict__ out, int* __restrict__ kv_len) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int S = L + num_token;
  int last = 0;
  for (int s = lane; s < S; s += 64) {
    const unsigned char m = s < num_token ? (unsigned char)0 : (in[(size_t)n * L + (s - num_token)] != 0 ? (unsigned char)1 : (unsigned char)0);
    out[(size_t)n * S + s] = m;
    if (m == 0) last = s + 1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(last, off, FF_WAVE);
    last = o > last ? o : last;
  }
  if (lane == 0) kv_len[n] = last;
}
extern "C" int ff_prepare_mask(const unsigned char* input_mask, int N, int L, int num_token, unsigned char* mask_out,
                               int* kv_len, ff_stream_t stream) {
  FF_CHECK_ARG(N > 0 && L >= 0 && num_token >= 0 && L + num_token > 0 && mask_out && kv_len && (L == 0 || input_mask),
               "ff_prepare_mask: bad arguments");
  hipLaunchKernelGGL(prepare_mask_kernel, dim3(ff_cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, input_mask, N, L, num_token,
                     mask_out, kv_len);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

// ---- LayerNorm affine folded into the following Linear ------------------------------------------------------
__global__ __launch_bounds__(256) void scale_columns_kernel(const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ gamma, float* __restrict__ out,
                                                            int N, int nvec) {
  const size_t total = (size_t)N * nvec;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / nvec), vi = (int)(i % nvec);
    const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)n * ldw + vi * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + vi * 4);
    *reinterpret_cast<f32x4*>(out + ((size_t)n * nvec + vi) * 4) = w * g;
  }
}

extern "C" int ff_fold_layernorm_linear(const float* W, int ldw, int N, int K, const float* bias, const float* gamma,
                                        const float* beta, const float* pos, int ldpos, int pos_rows, int pos_cols,
                                        float* Wf, float* bf, float* P, ff_stream_t stream) {
  FF_CHECK_ARG(W && gamma && beta && Wf && bf && N > 0 && K > 0 && (K & 3) == 0 && (ldw & 3) == 0 && ldw >= K,
               "ff_fold_layernorm_linear: bad arguments");
  FF_CHECK_ARG(ff_aligned16(W) && ff_aligned16(gamma) && ff_aligned16(beta) && ff_aligned16(Wf),
               "ff_fold_layernorm_linear: tensors must be 16-byte aligned");
  FF_CHECK_ARG(!pos || (P && pos_rows > 0 && pos_cols > 0 && pos_cols <= N && (ldpos & 3) == 0 && ldpos >= K &&
                        ff_aligned16(pos)), "ff_fold_layernorm_linear: bad position table arguments");
  hipStream_t st = (hipStream_t)stream;
  const size_t total = (size_t)N * (K / 4);
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_columns_kernel, dim3(grid), dim3(256), 0, st, W, ldw, gamma, Wf, N, K / 4);
  FF_CHECK_LAUNCH();
  // bf = beta W^T + bias: one-row product; P = pos W^T
  FF_RETURN_IF(ff_gemm_f32(beta, K, nullptr, 0, W, ldw, bias, nullptr, 0, bf, N, 1, N, K, 0, 1, stream));
  if (pos) FF_RETURN_IF(ff_gemm_f32(pos, ldpos, nullptr, 0, W, ldw, nullptr, nullptr, 0, P, pos_cols, pos_rows, pos_cols, K, 0, 1, stream));
  return FF_OK;
}
