// Shared helpers of libfaceformer_hip (gfx950 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "faceformer_hip.h"

#define FF_WAVE 64
#define FF_MAX_STREAMS 8

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- error reporting -------------------------------------------------------------------------
void ff_set_error(const char* fmt, ...);

#define FF_CHECK_ARG(cond, ...)           \
  do {                                    \
    if (!(cond)) {                        \
      ff_set_error(__VA_ARGS__);          \
      return FF_ERR_ARG;                  \
    }                                     \
  } while (0)

#define FF_CHECK_HIP(expr)                                                             \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      ff_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return FF_ERR_LAUNCH;                                                            \
    }                                                                                  \
  } while (0)

#define FF_CHECK_LAUNCH() FF_CHECK_HIP(hipGetLastError())

#define FF_RETURN_IF(expr)       \
  do {                           \
    int _s = (expr);             \
    if (_s != FF_OK) return _s;  \
  } while (0)

static inline bool ff_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ff_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t ff_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional per-category event profiling (bench.py's roofline leg) ---------------------------
// When enabled via ff_profile_begin(), every op launch is bracketed by a hipEvent pair on its own
// stream; ff_profile_end() synchronises and sums elapsed time / algorithmic work per category.
enum ff_prof_cat { FF_CAT_GEMM = 0, FF_CAT_ATTN = 1, FF_CAT_LN = 2, FF_CAT_POINTER = 3, FF_CAT_ROWOP = 4, FF_CAT_CHAIN = 5, FF_CAT_GEMM_X3 = 6, FF_NUM_CAT = 7 };
bool ff_prof_enabled();
void ff_prof_open(int cat, double work, hipStream_t st);
void ff_prof_close(hipStream_t st);
void ff_prof_add_bytes(int cat, double bytes);  // algorithmic bytes of the launch (operands read + results written once)
struct FFProfScope {
  hipStream_t st;
  bool on;
  FFProfScope(int cat, double work, hipStream_t s) : st(s), on(ff_prof_enabled()) {
    if (on) ff_prof_open(cat, work, st);
  }
  ~FFProfScope() {
    if (on) ff_prof_close(st);
  }
};

// bumped by every setter that changes how an operator is launched (ff_set_gemm_tuning, ff_set_attention_algo): kept for
// callers that cache launch decisions
unsigned long long ff_tuning_epoch();
void ff_tuning_changed();

// compute units of the current device (cached per device; 256 on an MI355X in SPX mode)
int ff_num_cus();

// ---- tuning knobs (round 6): ONE table, read from the environment once, settable per process through ff_set_tuning() ----------
// Every A/B switch of the library lives here: the name is the environment variable that initialises it, the value
// an int.  Defaults are the product; tests and tools flip them in-process (no child process needed) and the engine takes ONE
// snapshot of the ones that shape a decode at the top of ff_decode / ff_decode_workspace_bytes.
enum FFKnob {
  FF_K_L0_FOLD = 0,               // 1: the pointer launch leaves the appended rows' LayerNorm statistics (0: a LayerNorm launch)
  FF_K_POINTER_FOLD,              // 1: decoder.norm + project + pointer dot products as one GEMM for one-wireframe micro-batches
  FF_K_LAST_QKV_ONE_LAUNCH_ROWS,  // pruned last layer: one q|k|v launch up to this many active rows
  FF_K_PINNED_COUNTERS,           // host-mapped stop-counter slots a decode may use (<= the 65536 allocated)
  FF_K_DEBUG_TIMING,              // ff_decode prints host-side enqueue / wait times
  FF_K_DMA_MIN_ROWS, FF_K_DMA_MIN_ROWS_N512, FF_K_DMA_MIN_ROWS_WIDE,   // rows from which a projection takes the LDS-DMA f32 kernel
  FF_K_SK_HYBRID, FF_K_SK_HYBRID_FIX, FF_K_SK_HYBRID_MAXLEFT8, FF_K_SK_HYBRID_MINU, FF_K_SK_HYBRID_FORCE,
  FF_K_NO_PANEL,                  // 1: small-M launches never take gemm_panel_kernel
  FF_K_X3_SMALL_SPLIT,            // split kernel: K-pieces on idle CUs below one tile per CU
  FF_K_RK_SPLIT_OLD, FF_K_RK_SPLIT_YOUNG, FF_K_RK_PHASE, FF_K_RK_ROTATE,   // K/V-resident attention probes
  FF_K_X3_NEED_N1024, FF_K_X3_NEED_N512,   // split products: rows needed by the 1024- / 512-column projections, in quarters of x3_min_rows
  FF_K_X2H_ATTN,                  // 1: cross-attention launches whose descriptor carries fp16 planes take the 2 x fp16 kernel
  FF_K_COUNT
};
int ff_knob(int id);

// 2 x fp16 cross-attention (ff_attention_x2h.hip)
size_t ff_attention_planes_stride();
bool ff_attention_x2h_ok(const ff_attn_desc& d);
int ff_attention_x2h_launch(const ff_attn_desc& d, const void* planes, long long plane_stride, hipStream_t st);

// ff_pointer_argmax with the decode engine's stop-rule hand-over (ff_pointer.hip)
struct ff_pointer_sync {
  int* seen;        // [B] or null: count_eq counts a sequence's first eq_value only (FF_STOP_EACH_EOS)
  int* arrive;      // device int, zero before the launch: arrivals of the launch's sequences
  int* host_slot;   // device-visible address of a host-mapped pinned int: receives the launch's counter
  int host_which;   // 0: count_ge, 1: count_eq
  float* next_stats;  // [B, E/32, 2] or null: (mean, M2) per 32-column segment of the rows written to next_rows -- the
                      // LayerNorm statistics the folded layer-0 projection of the NEXT step consumes (ff_gemm_f32_ln)
  int logits_ready;   // 1: `logits` already holds the raw dot products (the engine's folded project + pointer GEMM): p is not read
};
int ff_pointer_argmax_sync(const float* p, int ldp, const float* memory, int S, int E, const unsigned char* mask,
                           const int* kv_len, const unsigned char* extra_mask, int ldextra, int B, int seqs_per_group,
                           int* next_tok, float* best, float* second, float* logits, int ldlogits, float* next_rows,
                           int ldnext, int* count_ge, int ge_bound, int* count_eq, int eq_value,
                           const ff_pointer_sync* sync, ff_stream_t stream);

// out[c, r] = in[r, c] for an [rows, cols] fp32 matrix (ff_rowops.hip; the engine's per-call transposes)
int ff_transpose(const float* in, int ld_in, int rows, int cols, float* out, int ld_out, hipStream_t st);

// partial-tile workspace of the 3 x bf16 kernel for (current device, stream): allocate now (ff_gemm_x3.hip)
extern "C" int ff_x3_prepare_stream(hipStream_t st);
// ... and the same area as scratch memory for another kernel of that stream (at most 24 MB)
int ff_stream_scratch(hipStream_t st, size_t bytes, float** out);

// ---- device helpers --------------------------------------------------------------------------
__device__ __forceinline__ float ff_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, FF_WAVE);
  return v;
}
__device__ __forceinline__ float ff_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, FF_WAVE));
  return v;
}

// Exchange between the two 32-lane halves of a wave without the LDS crossbar: v_permlane32_swap leaves (lower half's value, upper
// half's value) in every lane; `__shfl_xor(v, 32)` is a ds_bpermute round trip (>100 cycles, and a lgkmcnt event in the middle of
// hand-counted LDS reads).  Bitwise the results of v + __shfl_xor(v, 32) / fmaxf(v, __shfl_xor(v, 32)) (both are commutative).
__device__ __forceinline__ float ff_halves_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float ff_halves_max(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Bijective XCD-aware remap of a linear block id: blocks are dispatched round-robin over the 8
// XCDs (block b -> XCD b % 8); give every XCD a contiguous range of logical ids so that
// neighbouring tiles (which share operand panels) hit the same L2.  Speed only, never correctness.
__device__ __forceinline__ int ff_xcd_remap(int bid, int nblocks) {
  const int NX = 8;
  int xcd = bid % NX, idx = bid / NX;
  int q = nblocks / NX, r = nblocks % NX;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
