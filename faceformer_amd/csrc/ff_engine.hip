// Whole-path engine: encoder and greedy pointer decode, driven from C++ so that one call enqueues
// every kernel of a wireframe batch (the Python host makes ONE ffi call per batch, not ~2700).
//
// Data layout in HBM (all fp32 unless stated):
//   memory   [N, S, E]          encoder output (post final LayerNorm), S = L + num_token
//   kvc[l]   [N*S, 2E]          cross-attention K | V of decoder layer l, projected ONCE per batch
//                               (the reference re-projects them per step over F copies of memory)
//   x0       [T, Bc, E]         per micro-batch, position-major decoder input: row (j, b) is the
//                               embedding row selected at position j of sequence b (append only)
//   tok      [T, Bc] int32      tokens, same indexing
//   qkv0     [T, Bc, 3E]        layer-0 self-attention q|k|v, filled one position per step
//   x,y,yq,o [R, E], qkv [R,3E], h [R, FF]   scratch for R = t*Bc active rows (position-major, so
//                               the active rows of step t are the first t*Bc rows: every GEMM of
//                               the step is one dense [t*Bc, K] x [K, N] product)
// Sequences b of a micro-batch belong to wireframe w0 + b / F; nothing is replicated per sequence.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "ff_common.h"

namespace {

struct Bump {
  char* base;
  size_t size, off;
  bool ok;
  Bump(void* p, size_t n) : base((char*)p), size(n), off(0), ok(true) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = ff_align_up(count * sizeof(T), 256);
    if (off + bytes > size) { ok = false; off += bytes; return nullptr; }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
};

size_t bump_bytes(size_t count, size_t elem) { return ff_align_up(count * elem, 256); }

// ---- small kernels -----------------------------------------------------------------------------
// Start tokens of one micro-batch: sequence i belongs to wireframe i / Fc of the chunk and is its compact
// sequence f = f0 + i % Fc.
__global__ void init_tokens_kernel(int* tok, int Bc, int Fc, int f0, const int* num_input, int variant,
                                   int pad_tok, int sos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bc) return;
  if (variant == FF_PARALLEL) {
    // anchors = arange(F) per wireframe, WITHOUT the +num_token offset (reference quirk C-3,
    // model_para.py:201); rows >= num_input[w] start from token num_token-1 (model_para.py:204-205)
    const int f = f0 + i % Fc;
    tok[i] = f < num_input[i / Fc] ? f : pad_tok;
  } else {
    tok[i] = sos;
  }
}

// steps_done from the per-step counters (the reference's stop rules, model_para.py:232 / model.py:207-210)
// The counters are kept per (step, micro-batch) -- every pointer launch owns one, which it also publishes to the host
// (ff_pointer_count_block) -- and summed here into cnt_tot[step].
__global__ void steps_kernel(const int* __restrict__ cnt_ge, const int* __restrict__ cnt_eq, int nch, int variant, int N,
                             int steps_enqueued, int no_stop, int* __restrict__ cnt_tot, int* __restrict__ steps_done_out) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int* cnt = variant == FF_PARALLEL ? cnt_ge : cnt_eq;
  int steps = steps_enqueued, cum = 0;
  bool found = false;
  for (int s = 0; s < steps_enqueued; ++s) {
    int v = 0;
    for (int c = 0; c < nch; ++c) v += cnt[(size_t)s * nch + c];
    cnt_tot[s] = v;
    cum += v;
    if (!found && !no_stop && (variant == FF_PARALLEL ? v == 0 : cum == N)) { steps = s + 1; found = true; }
  }
  *steps_done_out = steps;
}

// predict[(w, fo), j] (int64) = token of the compact sequence that stands for output row fo of wireframe w,
// or 0 after the stop step.  With padding-anchor de-duplication every row fo >= num_input[w] is the ONE
// padding-anchor sequence stored at compact index num_input[w] (those rows are identical by construction:
// same start token, same memory, same mask; reference model_para.py:204-205).  One launch per micro-batch:
// it writes the rows whose compact sequence lives in [f0, f0 + Fc) of its wireframes.
__global__ void finalize_chunk_kernel(const int* __restrict__ tok_all, int Btot, int T, const int* __restrict__ steps_p,
                                      const int* __restrict__ num_input, int dedup, int F, int w0, int nw, int Fc,
                                      int f0, int b0, int64_t* __restrict__ predict, int* __restrict__ seq_of_row) {
  const int steps = *steps_p;
  const size_t total = (size_t)nw * F * T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % T);
    const int fo = (int)((i / T) % F), wl = (int)(i / ((size_t)T * F));
    int f = fo;
    if (dedup) { const int n = num_input[w0 + wl]; f = fo < n ? fo : n; }
    if (f < f0 || f >= f0 + Fc) continue;
    const int seq = b0 + wl * Fc + (f - f0);
    const size_t row = (size_t)(w0 + wl) * F + fo;
    predict[row * T + j] = (j <= steps) ? (int64_t)tok_all[(size_t)j * Btot + seq] : (int64_t)0;
    if (seq_of_row && j == 0) seq_of_row[row] = seq;
  }
}

// ---- host-side helpers ---------------------------------------------------------------------------
int check_model(const ff_model* m) {
  FF_CHECK_ARG(m != nullptr, "null model");
  FF_CHECK_ARG(m->E > 0 && m->H > 0 && m->E == m->H * FF_HEAD_DIM, "model: E=%d must equal H=%d * 64", m->E, m->H);
  FF_CHECK_ARG(m->FF > 0 && (m->FF & 3) == 0, "model: bad FF=%d", m->FF);
  FF_CHECK_ARG(m->num_enc_layers >= 0 && m->num_enc_layers <= FF_MAX_LAYERS && m->num_dec_layers > 0 &&
                   m->num_dec_layers <= FF_MAX_LAYERS, "model: bad layer counts");
  FF_CHECK_ARG(m->in_dim > 0 && (m->in_dim & 3) == 0, "model: in_dim=%d must be a multiple of 4", m->in_dim);
  FF_CHECK_ARG(m->num_token > 0, "model: num_token");
  FF_CHECK_ARG(m->split_kind == 0 || m->split_kind == 1, "model: split_kind=%d (0 = bf16 x 3 planes, 1 = fp16 x 2 planes)", m->split_kind);
  return FF_OK;
}

struct Linear { const float* w; const float* b; };

int gemm(const float* A, int lda, const float* A2, int n_split, const float* W, int ldw,
         const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N, int K, int act,
         hipStream_t st) {
  return ff_gemm_f32(A, lda, A2, n_split, W, ldw, bias, res, ldr, C, ldc, M, N, K, act, 0, st);
}

// Rows from which the 3 x bf16 kernel beats the f32 family, by output width (profiles/r04/gemm_x3_variants.txt, one MI355X:
// N = 1536: 92 vs 65 TF/s at 1024 rows; N = 1024: 122 vs 94 at 2048 (69 vs 80 at 1024); the N = 512 projections, K = 512 or
// 1024: 96 vs 72 / 119 vs 89 at 3072 (67 vs 79 / 84 vs 93 at 2048)): x3_min_rows is the threshold of the widest product, the
// others need 7/4 and 11/4 times as many rows.
inline bool x3_wins(const ff_decode_params* prm, int M, int N, int K, int lda) {
  if (prm->x3_min_rows <= 0 || (K % 32) != 0 || K < 64 || (N & 3) != 0) return false;
  if ((size_t)M * (size_t)lda >= ((size_t)1 << 30)) return false;   // the split kernel's 32-bit byte offsets (x3_check_common): f32 family instead
  const long need = (long)prm->x3_min_rows * (N >= 1536 ? 4 : (N >= 1024 ? ff_knob(FF_K_X3_NEED_N1024) : ff_knob(FF_K_X3_NEED_N512))) / 4;
  return M >= need;
}

// The same product through the 3 x bf16 kernel when the weight's planes are bound and the launch is in the
// range where it wins.
int gemm_or_x3(const ff_model* m, const ff_decode_params* prm, const void* planes, const float* A, int lda, const float* A2, int n_split,
               const float* W, int ldw, const float* bias, const float* res, int ldr, float* C, int ldc, int M,
               int N, int K, int act, hipStream_t st) {
  if (planes && x3_wins(prm, M, N, K, lda) && (!A2 || (n_split % 128) == 0))
    return m->split_kind == 1 ? ff_gemm_x2h(A, lda, A2, n_split, planes, bias, res, ldr, C, ldc, M, N, K, act, st)
                              : ff_gemm_x3(A, lda, A2, n_split, planes, bias, res, ldr, C, ldc, M, N, K, act, st);
  return gemm(A, lda, A2, n_split, W, ldw, bias, res, ldr, C, ldc, M, N, K, act, st);
}

// Scratch of one in-flight micro-batch (one set per stream): R = t*Bc active rows, position-major.
struct Scratch {
  float *x, *y, *yq, *qkv, *o, *h, *p, *logits;
  float* lnstat;   // [R, E/32, 2] per-row segment statistics of x (FF_FUSE_LAYERNORM)
};

struct DecodeBuffers {
  float *mem_pos, *kvc[FF_MAX_LAYERS];
  unsigned char* kvp[FF_MAX_LAYERS];   // fp16 planes of the cross-attention K | V (2 x fp16 attention kernel) or null
  float *x0_all, *qkv0_all;
  float* x0stat_all;   // [Btot, E/32, 2] LayerNorm segment statistics of the NEWEST x0 rows (written by the pointer launches)
  float *projT, *pg_all, *pc_all;   // folded project + pointer GEMM of one-wireframe micro-batches (see pointer_fold): the transposed
                                    // folded project weight [E, E]; per micro-batch G = memory_w W' [S, E] and c = memory_w b' [S4]
  int* tok_all;   // [T, Btot] global, position-major
  Scratch scr[FF_MAX_STREAMS];
  int *cnt_ge, *cnt_eq;   // [T, nch] per (step, micro-batch)
  int *arrive;            // [T, nch] arrivals of the pointer launches (counter hand-over to the host)
  int *seen;              // [Btot] FF_STOP_EACH_EOS: the sequence has produced an EOS
  int *cnt_tot;           // [T] per-step totals (steps_kernel)
  int *steps_dev;
};

// A micro-batch is a contiguous range [b0, b0 + Bc) of the COMPACT sequence index: nw >= 1 consecutive
// wireframes with Fc sequences each (compact sequences [f0, f0 + Fc) of every one of them).  Without
// padding-anchor de-duplication the compact width of every wireframe is F and b = w*F + f as in the reference.
struct Chunk {
  int w0, nw, Fc, f0, b0, Bc, sid;
  float* x0;     // [T, Bc, E]
  float* qkv0;   // [T, Bc, 3E] or null
  float* x0stat; // [Bc, E/32, 2] statistics of the rows the last pointer launch appended to x0, or null
  float *pg, *pc; // one-wireframe micro-batch with the folded forms bound: G [S, E] and c [S] (logits = LN(x) G^T + c), or null
};

// Compact width of wireframe w: its num_input real anchors plus ONE padding-anchor sequence when it has fewer
// than F (the reference decodes F - num_input identical copies of it, model_para.py:204-205).
inline int compact_width(const ff_decode_params* p, const int* num_input_host, int w) {
  if (p->variant != FF_PARALLEL || !(p->flags & FF_DEDUP_PAD_ANCHORS) || !num_input_host) return p->F;
  int n = num_input_host[w];
  n = n < 0 ? 0 : n;
  return n < p->F ? n + 1 : p->F;
}

// Micro-batches: consecutive wireframes whose compact widths are within 25 % of the widest one share a chunk
// (its Fc = the widest; the narrower ones carry a few surplus padding-anchor copies), at most
// chunk_wireframes of them and at most chunk_max_seqs sequences; chunk_seqs > 0 additionally cuts every
// wireframe into sequence groups.  Callers that want tight chunks pass the wireframes sorted by edge count
// (the Python model does).
void plan_chunks(const ff_decode_params* p, const int* num_input_host, int ns, std::vector<Chunk>* out, int* btot,
                 int* max_bc, int* nchunks = nullptr) {
  const int N = p->N;
  int cw_lim = (p->chunk_wireframes <= 0 || p->chunk_wireframes > N) ? N : p->chunk_wireframes;
  // The single-sequence model decodes ONE sequence per wireframe (reference model.py:193-210: N sequences per step), so a
  // micro-batch of chunk_wireframes = 16 wireframes is 16 rows per position -- 64 wireframes then are 4 x 258 steps of
  // <= 4 128-row launches.  There the micro-batch is cut by SEQUENCES: up to chunk_max_seqs of them (0: chunk_wireframes as
  // before).  The cumulative EOS rule is untouched: every micro-batch adds to the same per-step counters.
  if (p->variant == FF_SEQ2SEQ && p->chunk_max_seqs > 0) cw_lim = p->chunk_max_seqs < N ? p->chunk_max_seqs : N;
  int b0 = 0, mx = 0, nc = 0;
  int w = 0;
  while (w < N) {
    int Fm = compact_width(p, num_input_host, w), Fmin = Fm, nw = 1;
    const bool split = p->chunk_seqs > 0 && p->chunk_seqs < Fm;
    while (!split && w + nw < N && nw < cw_lim) {
      const int c = compact_width(p, num_input_host, w + nw);
      const int nmax = c > Fm ? c : Fm, nmin = c < Fmin ? c : Fmin;
      if (4 * nmin < 3 * nmax) break;
      if (p->chunk_max_seqs > 0 && (long)(nw + 1) * nmax > (long)(p->chunk_max_seqs > nmax ? p->chunk_max_seqs : nmax)) break;
      Fm = nmax; Fmin = nmin; ++nw;
    }
    const int fstep = split ? p->chunk_seqs : Fm;
    for (int f0 = 0; f0 < Fm; f0 += fstep) {
      Chunk c;
      c.w0 = w; c.nw = nw; c.f0 = f0;
      c.Fc = (Fm - f0) < fstep ? (Fm - f0) : fstep;
      c.b0 = b0; c.Bc = nw * c.Fc;
      c.sid = (int)(out ? out->size() % (size_t)ns : 0);
      c.x0 = nullptr; c.qkv0 = nullptr; c.x0stat = nullptr; c.pg = nullptr; c.pc = nullptr;
      b0 += c.Bc;
      mx = c.Bc > mx ? c.Bc : mx;
      ++nc;
      if (out) out->push_back(c);
    }
    w += nw;
  }
  *btot = b0;
  *max_bc = mx;
  if (nchunks) *nchunks = nc;
}

int plan_streams(const ff_decode_params* p) {
  return p->num_streams < 1 ? 1 : (p->num_streams > FF_MAX_STREAMS ? FF_MAX_STREAMS : p->num_streams);
}

// The tuning knobs that shape a decode, read ONCE per ff_decode / ff_decode_workspace_bytes call: the workspace
// layout and every step of that call see the same values whatever ff_set_tuning() does meanwhile.  The two that change the
// LAYOUT can also be switched off per call through ff_decode_params.flags (FF_NO_L0_FOLD, FF_NO_POINTER_FOLD).
constexpr int FF_PINNED_SLOTS = 65536;   // host-mapped stop counters allocated per device (knob FF_PINNED_COUNTERS: how many a decode may use)
struct EngineKnobs {
  bool l0_fold, pointer_fold, dbg_timing;
  bool x2h_attn;             // cross-attention K | V also as fp16 planes (FF_X2H_ATTN, with the fp16 split products in use)
  int one_launch_rows, pinned;
};
EngineKnobs engine_knobs(const ff_decode_params* p) {
  EngineKnobs k;
  k.l0_fold = ff_knob(FF_K_L0_FOLD) != 0 && !(p->flags & FF_NO_L0_FOLD);
  k.pointer_fold = ff_knob(FF_K_POINTER_FOLD) != 0 && !(p->flags & FF_NO_POINTER_FOLD);
  k.dbg_timing = ff_knob(FF_K_DEBUG_TIMING) != 0;
  k.x2h_attn = ff_knob(FF_K_X2H_ATTN) != 0;
  k.one_launch_rows = ff_knob(FF_K_LAST_QKV_ONE_LAUNCH_ROWS);
  const int pc = ff_knob(FF_K_PINNED_COUNTERS);
  k.pinned = pc > 0 && pc < FF_PINNED_SLOTS ? pc : FF_PINNED_SLOTS;
  return k;
}

// Workspace layout for `btot` compact sequences in micro-batches of at most `max_bc`.
bool can_fuse_layernorm(const ff_model* m, const ff_decode_params* prm);

size_t layout_decode(const ff_model* m, const ff_decode_params* p, const EngineKnobs& kn, size_t Btot, size_t Bch, size_t nch,
                     Bump& bp, DecodeBuffers* out) {
  const int E = m->E, FFd = m->FF, S = p->L + m->num_token, T = p->T;
  const int ns = plan_streams(p);
  const size_t Rmax = (size_t)(T - 1 > 0 ? T - 1 : 1) * Bch;
  DecodeBuffers b;
  memset(&b, 0, sizeof(b));
  b.mem_pos = bp.take<float>((size_t)p->N * S * E);
  for (int l = 0; l < m->num_dec_layers; ++l) b.kvc[l] = bp.take<float>((size_t)p->N * S * 2 * E);
  // the package default's cross-attention runs on the fp16 matrix cores as well (ff_attention_x2h.hip): K | V of every (wireframe,
  // head) pair split once per batch into fp16 planes, 145 KB per pair and layer (key sets of at most 288 rows)
  const bool planes = kn.x2h_attn && m->split_kind == 1 && p->x3_min_rows > 0 && S <= 288 && m->dec[0].in_proj_planes != nullptr;
  for (int l = 0; l < m->num_dec_layers; ++l)
    b.kvp[l] = planes ? bp.take<unsigned char>(ff_attention_planes_bytes(p->N, m->H)) : nullptr;
  b.x0_all = bp.take<float>((size_t)T * Btot * E);
  b.tok_all = bp.take<int>((size_t)T * Btot);
  b.qkv0_all = (p->flags & FF_REUSE_LAYER0_QKV) ? bp.take<float>((size_t)T * Btot * 3 * E) : nullptr;
  // (FF_L0_FOLD=0 / FF_NO_L0_FOLD: the newest rows' LayerNorm as its own launch, as before round 5 -- A/B runs and tests of that form)
  b.x0stat_all = (kn.l0_fold && (p->flags & FF_REUSE_LAYER0_QKV) && can_fuse_layernorm(m, p)) ? bp.take<float>(Btot * (size_t)(E / 32) * 2)
                                                                                 : nullptr;   // (size query: take() returns null)
  // (FF_POINTER_FOLD=0 / FF_NO_POINTER_FOLD: project and the pointer GEMM as two launches, as before round 5)
  const bool pf = kn.pointer_fold && can_fuse_layernorm(m, p);
  b.projT = pf ? bp.take<float>((size_t)E * E) : nullptr;
  b.pg_all = pf ? bp.take<float>(nch * (size_t)S * E) : nullptr;
  b.pc_all = pf ? bp.take<float>(nch * (size_t)((S + 3) & ~3)) : nullptr;
  for (int s = 0; s < ns; ++s) {
    Scratch& c = b.scr[s];
    c.x = bp.take<float>(Rmax * E);
    c.y = bp.take<float>(Rmax * E);
    c.yq = bp.take<float>(Rmax * E);
    c.qkv = bp.take<float>(Rmax * 3 * E);
    c.o = bp.take<float>(Rmax * E);
    c.h = bp.take<float>(Rmax * FFd);
    c.p = bp.take<float>(Bch * E);
    c.logits = bp.take<float>(Bch * (size_t)S);
    c.lnstat = bp.take<float>(Rmax * (size_t)(E / 32 + 1) * 2);
  }
  b.cnt_ge = bp.take<int>((size_t)T * nch);   // (these four stay in this order, back to back: ff_decode zeroes them with one fill)
  b.cnt_eq = bp.take<int>((size_t)T * nch);
  b.arrive = bp.take<int>((size_t)T * nch);
  b.seen = bp.take<int>(Btot);
  b.cnt_tot = bp.take<int>(T);
  b.steps_dev = bp.take<int>(4);
  if (out) *out = b;
  return bp.off;
}

// LayerNorm fusion is possible when the folded weights are bound and the shapes fit the fused GEMM forms.
bool can_fuse_layernorm(const ff_model* m, const ff_decode_params* prm) {
  if (!(prm->flags & FF_FUSE_LAYERNORM)) return false;
  if (m->E % 64 != 0 || m->E < 128 || m->E > 512 || m->FF % 64 != 0 || m->FF < 128) return false;
  if (!m->proj_fold_w || !m->proj_fold_b) return false;
  for (int l = 0; l < m->num_dec_layers; ++l) {
    const ff_layer_weights& w = m->dec[l];
    if (!w.ln1_w || !w.ln1_b || !w.ln1_pos || !w.ln2_w || !w.ln2_b || !w.ln2_pos || !w.ln3_w || !w.ln3_b) return false;
  }
  return true;
}

// Does a decode step with R active rows take the LayerNorm-folded projections?  (decoder_pass and the engine loop ask.)
bool step_fuses(const ff_model* m, const ff_decode_params* prm, long R) {
  const int nd = m->num_dec_layers, E = m->E;
  const bool x3_bound = prm->x3_min_rows > 0 && nd > 0 && m->dec[0].in_proj_planes != nullptr;
  const bool x3_folds = x3_bound && m->dec[0].ln1_planes != nullptr && m->dec[0].ln2_planes != nullptr &&
                        m->dec[0].ln3_planes != nullptr && E == 512;
  // f32 only: since round 4 the LDS-DMA kernel of the f32 family carries the folded forms at every size too (K = E = 512), so
  // the steps fold at every size as well (128 wireframes per call: 5 053 -> 301 LayerNorm launches, 206 -> 214 k selections/s).
  const int fuse_max = prm->ln_fuse_max_rows > 0 ? prm->ln_fuse_max_rows
                       : ((x3_folds || (!x3_bound && E == 512)) ? (1 << 30)
                          : (x3_bound && prm->x3_min_rows - 1 < 12288 ? prm->x3_min_rows - 1 : 12288));
  return can_fuse_layernorm(m, prm) && R <= fuse_max;
}

// One decoder pass over the current prefix (t positions) of one micro-batch.
// full_rows: evaluate every layer for all rows and project all rows into proj_all (ld = E rows
// position-major within the chunk); otherwise the result is p[Bc, E] for the newest position.
//
// With FF_FUSE_LAYERNORM only layer 0's norm1 is a standalone launch: every other LayerNorm input x is produced
// by a projection with a residual (out-proj, linear2), which leaves per-row segment statistics in `lnstat`; the
// projection that consumes LN(x) (+ qpos) reads x and the statistics and applies gamma / beta / qpos W^T through
// folded weights (ff_gemm_f32_ln).  19 -> 1 LayerNorm launches per decode step of a 6-layer decoder.
int decoder_pass(const ff_model* m, const ff_decode_params* prm, const EngineKnobs& kn, const DecodeBuffers& bufs, const Scratch& buf,
                 const Chunk& ck, const unsigned char* mask, const int* kv_len, int t, bool full_rows,
                 float* proj_all, hipStream_t st, float* logits_out = nullptr) {
  const int E = m->E, FFd = m->FF, H = m->H, S = prm->L + m->num_token, F = ck.Fc, T = prm->T;
  const int Bc = ck.Bc, R = t * Bc, nd = m->num_dec_layers;
  const size_t newoff = (size_t)(t - 1) * Bc;
  const bool reuse0 = (prm->flags & FF_REUSE_LAYER0_QKV) != 0 && ck.qkv0 != nullptr;
  const bool prune_last = (prm->flags & FF_LAST_LAYER_LAST_ROW) != 0 && !full_rows;
  // Folding the LayerNorms into the projections removes 18 launches per step.  On MI355X the normalising consumer costs about
  // what the standalone LayerNorm launch it replaces costs up to ~10^4 rows (round 3, once the position-table term of the
  // epilogue became ONE load per lane: config B 62.1-62.4 ms folded at every step vs 62.6 with the round-2 limit of 4096 rows);
  // above that the plain launches take the 128x64-tile kernel, which the fused forms do not have (config C / E micro-batches),
  // so the fused form is used up to ln_fuse_max_rows active rows (default 12288) -- both forms are parity-tested.
  // With the 3 x bf16 projections bound, the steps that take them (x3_min_rows on) launch the LayerNorms: the split kernel
  // has no folded form, and LayerNorm + split product beats the folded f32 forms there (config B 60.1 vs 61.9 ms).
  // Round 4: the 3 x bf16 kernel has the folded forms as well (ff_gemm_x3_ln).  With the planes of the folded weights bound the
  // steps fold at EVERY size (the large launches then take the split kernel, which has no 128x64-tile problem).
  const bool fuse = step_fuses(m, prm, R);
  const int nseg = E / 32;
  const float* qpos = m->qpos_table;
  const float* qpos_new = qpos + (size_t)(t - 1) * E;

  // C = act(LN?(A) W^T + bias [+ table]) [+ residual], optionally leaving the row statistics of C
  // `planes` (optional): the bf16 planes of the [plane_rows, K] weight whose rows [row0, row0 + N) are W
  auto gemm_ln = [&](const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldr,
                     float* C, int ldc, int M, int N, int K, int act, const float* st_in, const float* table, int ldt,
                     int tcols, float* st_out, const void* planes = nullptr, int plane_rows = 0, int row0 = 0,
                     const float* colsum = nullptr) -> int {
    ff_gemm_ln_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.lda = lda; d.W = W; d.ldw = ldw; d.bias = bias; d.residual = res; d.ldr = ldr; d.C = C; d.ldc = ldc;
    d.M = M; d.N = N; d.K = K; d.act = act; d.tile = 0;
    d.ln_stats_in = st_in; d.ln_nseg = K / 32; d.ln_eps = m->ln_eps;
    d.row_table = table; d.ld_row_table = ldt; d.row_div = Bc; d.row_cols = tcols;
    d.ln_stats_out = st_out;
    if (planes && x3_wins(prm, M, N, K, lda) && (!st_in || K == 512) && (!table || (tcols & 3) == 0))
      return m->split_kind == 1 ? ff_gemm_x2h_ln(&d, planes, plane_rows, row0, st_in ? colsum : nullptr, st)
                                : ff_gemm_x3_ln(&d, planes, plane_rows, row0, st_in ? colsum : nullptr, st);
    return ff_gemm_f32_ln(&d, st);
  };

  // the LayerNorm-folded projection over all R rows that opens layer l2 > 0: q|k|v, or k|v alone when the layer is pruned to
  // its newest position (its q then covers Bc rows only)
  // The pruned last layer needs k | v of every row and q of the newest position only: two launches.  On launch-bound steps
  // (few rows) ONE q | k | v launch over all rows is cheaper than the second launch it saves (FF_LAST_QKV_ONE_LAUNCH_ROWS: up
  // to this many active rows; 0 = never); the q of the older rows is computed and not used.
  const bool last_qkv_one = R <= kn.one_launch_rows;
  auto first_proj = [&](int l2) -> int {
    const ff_layer_weights& w2 = m->dec[l2];
    if (prune_last && l2 == nd - 1 && t > 1 && !last_qkv_one)
      return gemm_ln(buf.x, E, w2.ln1_w + (size_t)E * E, E, w2.ln1_b + E, nullptr, 0, buf.qkv + E, 3 * E, R, 2 * E, E, 0,
                     buf.lnstat, w2.ln1_pos + E, 2 * E, E, nullptr, w2.ln1_planes, 3 * E, E, w2.ln1_csum);
    return gemm_ln(buf.x, E, w2.ln1_w, E, w2.ln1_b, nullptr, 0, buf.qkv, 3 * E, R, 3 * E, E, 0, buf.lnstat, w2.ln1_pos, 2 * E,
                   2 * E, nullptr, w2.ln1_planes, 3 * E, 0, w2.ln1_csum);
  };
  for (int l = 0; l < nd; ++l) {
    const ff_layer_weights& w = m->dec[l];
    const bool last = prune_last && (l == nd - 1);
    const float* xin = (l == 0) ? ck.x0 : buf.x;
    const float* QKV;
    // ---- self attention: q = k = LN1(x) + qpos, v = LN1(x), no mask (transformer.py:242-246) ----
    if (l == 0 && reuse0) {
      if (full_rows) {
        // (the pass behind the loop: every position < t went through a decode step, its layer-0 q|k|v is in the cache)
      } else if (fuse && t > 1 && ck.x0stat) {
        // the newest rows were appended by the previous step's pointer launch together with their segment statistics: the
        // folded projection normalises them itself (no LayerNorm launch left in a decode step after the first)
        FF_RETURN_IF(gemm_ln(xin + newoff * E, E, w.ln1_w, E, w.ln1_b, nullptr, 0, ck.qkv0 + newoff * 3 * E, 3 * E, Bc, 3 * E, E,
                             0, ck.x0stat, w.ln1_pos + (size_t)(t - 1) * 2 * E, 2 * E, 2 * E, nullptr));
      } else {
        FF_RETURN_IF(ff_layernorm(xin + newoff * E, E, w.norm1_w, w.norm1_b, m->ln_eps, buf.y, E, buf.yq, E,
                                  qpos_new, E, Bc, 1, Bc, E, st));
        FF_RETURN_IF(gemm(buf.yq, E, buf.y, 2 * E, w.self_attn.in_proj_w, E, w.self_attn.in_proj_b, nullptr, 0,
                          ck.qkv0 + newoff * 3 * E, 3 * E, Bc, 3 * E, E, 0, st));
      }
      QKV = ck.qkv0;
    } else if (fuse && l > 0) {
      FF_RETURN_IF(first_proj(l));
      if (last && t > 1 && !last_qkv_one) {
        // the pruned last layer attends from its newest position only: k | v for every row (above), q for the last Bc rows
        FF_RETURN_IF(gemm_ln(xin + newoff * E, E, w.ln1_w, E, w.ln1_b, nullptr, 0, buf.qkv + newoff * 3 * E, 3 * E, Bc, E, E,
                             0, buf.lnstat + newoff * nseg * 2, w.ln1_pos + (size_t)(t - 1) * 2 * E, 2 * E, E, nullptr));
      }
      QKV = buf.qkv;
    } else {
      FF_RETURN_IF(ff_layernorm(xin, E, w.norm1_w, w.norm1_b, m->ln_eps, buf.y, E, buf.yq, E, qpos, E, Bc, T,
                                R, E, st));
      const long x3_need = (long)prm->x3_min_rows;   // (the bf16 planes cover the whole [3E, E] weight: no row ranges)
      const bool x3_here = w.in_proj_planes && prm->x3_min_rows > 0 && R >= x3_need;
      if (last && t > 1 && !x3_here && (E % 64) == 0) {
        // pruned last layer: k (from LN(x)+qpos) | v (from LN(x)) for every row, q for the newest position only
        FF_RETURN_IF(gemm(buf.yq, E, buf.y, E, w.self_attn.in_proj_w + (size_t)E * E, E, w.self_attn.in_proj_b + E, nullptr,
                          0, buf.qkv + E, 3 * E, R, 2 * E, E, 0, st));
        FF_RETURN_IF(gemm(buf.yq + newoff * E, E, nullptr, 0, w.self_attn.in_proj_w, E, w.self_attn.in_proj_b, nullptr, 0,
                          buf.qkv + newoff * 3 * E, 3 * E, Bc, E, E, 0, st));
      } else {
        FF_RETURN_IF(gemm_or_x3(m, prm, w.in_proj_planes, buf.yq, E, buf.y, 2 * E, w.self_attn.in_proj_w, E,
                                w.self_attn.in_proj_b, nullptr, 0, buf.qkv, 3 * E, R, 3 * E, E, 0, st));
      }
      QKV = buf.qkv;
    }
    // rows that continue through the rest of this layer
    const size_t roff = last ? newoff : 0;
    const int Rl = last ? Bc : R;
    float* stat = buf.lnstat + roff * nseg * 2;
    {
      ff_attn_desc d;
      memset(&d, 0, sizeof(d));
      d.q = QKV + roff * 3 * E;  d.ldq = 3 * E;
      d.k = QKV + E;             d.ldk = 3 * E;
      d.v = QKV + 2 * E;         d.ldv = 3 * E;
      d.o = buf.o + roff * E;    d.ldo = E;
      d.num_groups = Bc; d.num_heads = H;
      d.nq = last ? 1 : t;
      d.q_group_stride = 1; d.q_inner = 1; d.q_outer_stride = Bc;
      d.nk = t; d.k_group_stride = 1; d.k_stride = Bc;
      d.scale = 0.125f;
      FF_RETURN_IF(ff_attention(&d, st));
    }
    float* qc = buf.qkv;  // [rows, E] view of the scratch
    if (fuse) {
      FF_RETURN_IF(gemm_ln(buf.o + roff * E, E, w.self_attn.out_w, E, w.self_attn.out_b, xin + roff * E, E,
                           buf.x + roff * E, E, Rl, E, E, 0, nullptr, nullptr, 0, 0, stat, w.self_out_planes, E, 0));
      // ---- cross attention: q = LN2(x) + qpos (transformer.py:247-252) ----
      FF_RETURN_IF(gemm_ln(buf.x + roff * E, E, w.ln2_w, E, w.ln2_b, nullptr, 0, qc + roff * E, E, Rl, E, E, 0, stat,
                           w.ln2_pos + (last ? (size_t)(t - 1) * E : 0), E, E, nullptr, w.ln2_planes, E, 0, w.ln2_csum));
    } else {
      FF_RETURN_IF(gemm_or_x3(m, prm, w.self_out_planes, buf.o + roff * E, E, nullptr, 0, w.self_attn.out_w, E,
                              w.self_attn.out_b, xin + roff * E, E, buf.x + roff * E, E, Rl, E, E, 0, st));
      // ---- cross attention: q = LN2(x) + qpos, k = memory + pos, v = memory (transformer.py:247-252);
      //      K/V come from the per-batch cache ----
      if (last)
        FF_RETURN_IF(ff_layernorm(buf.x + roff * E, E, w.norm2_w, w.norm2_b, m->ln_eps, nullptr, 0, buf.yq + roff * E,
                                  E, qpos_new, E, Bc, 1, Rl, E, st));
      else
        FF_RETURN_IF(ff_layernorm(buf.x, E, w.norm2_w, w.norm2_b, m->ln_eps, nullptr, 0, buf.yq, E, qpos, E, Bc, T,
                                  Rl, E, st));
      FF_RETURN_IF(gemm_or_x3(m, prm, w.cross_q_planes, buf.yq + roff * E, E, nullptr, 0, w.cross_attn.in_proj_w, E,
                              w.cross_attn.in_proj_b, nullptr, 0, qc + roff * E, E, Rl, E, E, 0, st));
    }
    {
      ff_attn_desc d;
      memset(&d, 0, sizeof(d));
      d.q = qc + roff * E;      d.ldq = E;
      d.k = bufs.kvc[l] + (size_t)ck.w0 * S * 2 * E;     d.ldk = 2 * E;
      d.v = d.k + E;            d.ldv = 2 * E;
      d.o = buf.o + roff * E;   d.ldo = E;
      d.num_groups = ck.nw; d.num_heads = H;
      d.nq = last ? F : F * t;
      d.q_group_stride = F; d.q_inner = F; d.q_outer_stride = Bc;
      d.nk = S; d.k_group_stride = S; d.k_stride = 1;
      d.kv_len = kv_len + ck.w0;
      d.key_mask = mask + (size_t)ck.w0 * S; d.mask_stride = S;
      d.scale = 0.125f;
      if (bufs.kvp[l] && x3_wins(prm, R, 3 * E, E, E))   // (the steps whose projections take the split products)
        d.kv_planes = bufs.kvp[l] + (size_t)ck.w0 * H * (ff_attention_planes_bytes(1, H) / H);
      FF_RETURN_IF(ff_attention(&d, st));
    }
    if (fuse) {
      FF_RETURN_IF(gemm_ln(buf.o + roff * E, E, w.cross_attn.out_w, E, w.cross_attn.out_b, buf.x + roff * E, E,
                           buf.x + roff * E, E, Rl, E, E, 0, nullptr, nullptr, 0, 0, stat, w.cross_out_planes, E, 0));
      // ---- feed forward (transformer.py:253-255) ----
      FF_RETURN_IF(gemm_ln(buf.x + roff * E, E, w.ln3_w, E, w.ln3_b, nullptr, 0, buf.h + roff * FFd, FFd, Rl, FFd, E, 1,
                           stat, nullptr, 0, 0, nullptr, w.ln3_planes, FFd, 0, w.ln3_csum));
      FF_RETURN_IF(gemm_ln(buf.h + roff * FFd, FFd, w.lin2_w, FFd, w.lin2_b, buf.x + roff * E, E, buf.x + roff * E, E,
                           Rl, E, FFd, 0, nullptr, nullptr, 0, 0, stat, w.lin2_planes, E, 0));
    } else {
      FF_RETURN_IF(gemm_or_x3(m, prm, w.cross_out_planes, buf.o + roff * E, E, nullptr, 0, w.cross_attn.out_w, E,
                              w.cross_attn.out_b, buf.x + roff * E, E, buf.x + roff * E, E, Rl, E, E, 0, st));
      // ---- feed forward (transformer.py:253-255) ----
      FF_RETURN_IF(ff_layernorm(buf.x + roff * E, E, w.norm3_w, w.norm3_b, m->ln_eps, buf.y + roff * E, E, nullptr, 0,
                                nullptr, 0, 1, 1, Rl, E, st));
      FF_RETURN_IF(gemm_or_x3(m, prm, w.lin1_planes, buf.y + roff * E, E, nullptr, 0, w.lin1_w, E, w.lin1_b, nullptr, 0,
                              buf.h + roff * FFd, FFd, Rl, FFd, E, 1, st));
      FF_RETURN_IF(gemm_or_x3(m, prm, w.lin2_planes, buf.h + roff * FFd, FFd, nullptr, 0, w.lin2_w, FFd, w.lin2_b,
                              buf.x + roff * E, E, buf.x + roff * E, E, Rl, E, FFd, 0, st));
    }
  }
  // ---- decoder.norm + project (transformer.py:115-116, model_para.py:225) ----
  if (fuse) {
    if (full_rows)
      FF_RETURN_IF(gemm_ln(buf.x, E, m->proj_fold_w, E, m->proj_fold_b, nullptr, 0, proj_all, E, R, E, E, 0, buf.lnstat,
                           nullptr, 0, 0, nullptr));
    else if (logits_out && ck.pg)
      // pointer_fold: logits = <project(LN(x)), memory_s> = LN(x) (memory W')^T + memory b' -- ONE launch for decoder.norm,
      // project and the pointer's dot products of a one-wireframe micro-batch (G and c are made once per call)
      FF_RETURN_IF(gemm_ln(buf.x + newoff * E, E, ck.pg, E, ck.pc, nullptr, 0, logits_out, S, Bc, S, E, 0,
                           buf.lnstat + newoff * nseg * 2, nullptr, 0, 0, nullptr));
    else
      FF_RETURN_IF(gemm_ln(buf.x + newoff * E, E, m->proj_fold_w, E, m->proj_fold_b, nullptr, 0, buf.p, E, Bc, E, E, 0,
                           buf.lnstat + newoff * nseg * 2, nullptr, 0, 0, nullptr));
  } else if (full_rows) {
    FF_RETURN_IF(ff_layernorm(buf.x, E, m->dec_norm_w, m->dec_norm_b, m->ln_eps, buf.y, E, nullptr, 0, nullptr, 0,
                              1, 1, R, E, st));
    FF_RETURN_IF(gemm(buf.y, E, nullptr, 0, m->proj_w, E, m->proj_b, nullptr, 0, proj_all, E, R, E, E, 0, st));
  } else {
    FF_RETURN_IF(ff_layernorm(buf.x + newoff * E, E, m->dec_norm_w, m->dec_norm_b, m->ln_eps, buf.y, E, nullptr, 0,
                              nullptr, 0, 1, 1, Bc, E, st));
    FF_RETURN_IF(gemm(buf.y, E, nullptr, 0, m->proj_w, E, m->proj_b, nullptr, 0, buf.p, E, Bc, E, E, 0, st));
  }
  return FF_OK;
}

// Internal side streams + fork/join events: one pool per device, created on first use.
// host-mapped ints: one per (step, micro-batch) of a decode; a decode with more of them checks its stop rule by draining the
// streams and copying (FF_PINNED_COUNTERS overrides the size: tests run that path with a handful of slots)
// (knob FF_PINNED_COUNTERS: how many of them a decode may use)
struct StreamPool {
  hipStream_t side[FF_MAX_STREAMS];
  hipEvent_t fork_ev, join_ev[FF_MAX_STREAMS];
  hipEvent_t chk_ev[FF_MAX_STREAMS];      // stop-rule check: per-stream progress marks
  int* hpin;                                // host-mapped pinned counters [step][micro-batch], written by the pointer launches
  int* hpin_dev;                            // ... the device-visible address of the same memory
  int created;
  bool events;
};
constexpr int FF_MAX_DEVICES = 16;
StreamPool g_pools[FF_MAX_DEVICES];
std::mutex g_pool_mu;
// The pool's side streams, events and pinned counters belong to ONE decode at a time: host threads that
// decode on the same device take turns (different devices run concurrently).
std::mutex g_pool_busy[FF_MAX_DEVICES];

int pool_get(int n, StreamPool** out) {
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  FF_CHECK_ARG(dev >= 0 && dev < FF_MAX_DEVICES, "device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_pool_mu);
  StreamPool& pool = g_pools[dev];
  if (!pool.events) {
    FF_CHECK_HIP(hipEventCreateWithFlags(&pool.fork_ev, hipEventDisableTiming));
    for (int i = 0; i < FF_MAX_STREAMS; ++i) {
      FF_CHECK_HIP(hipEventCreateWithFlags(&pool.join_ev[i], hipEventDisableTiming));
      FF_CHECK_HIP(hipEventCreateWithFlags(&pool.chk_ev[i], hipEventDisableTiming));
    }
    // coherent (fine-grained) host memory mapped into the device's address space: the pointer launches store their stop-rule
    // counters straight into it (system-scope stores; no copy launch between two decode steps)
    FF_CHECK_HIP(hipHostMalloc(reinterpret_cast<void**>(&pool.hpin), sizeof(int) * FF_PINNED_SLOTS,
                               hipHostMallocMapped | hipHostMallocCoherent));
    FF_CHECK_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&pool.hpin_dev), pool.hpin, 0));
    pool.events = true;
  }
  while (pool.created < n) {
    FF_CHECK_HIP(hipStreamCreateWithFlags(&pool.side[pool.created], hipStreamNonBlocking));
    FF_RETURN_IF(ff_gemm_prepare_stream(pool.side[pool.created]));
    pool.created++;
  }
  *out = &pool;
  return FF_OK;
}

std::mutex* pool_busy_mutex() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FF_MAX_DEVICES) return nullptr;
  return &g_pool_busy[dev];
}

}  // namespace

// =================================================================================================
extern "C" size_t ff_encode_workspace_bytes(const ff_model* m, int N, int L) {
  if (!m || N <= 0 || L < 0) return 0;
  const size_t S = (size_t)L + m->num_token, E = m->E;
  size_t tot = 0;
  tot += 2 * bump_bytes((size_t)N * L * E, 4);          // embedding MLP hidden / output
  tot += 4 * bump_bytes((size_t)N * S * E, 4);          // x, y, yq, o
  tot += bump_bytes((size_t)N * S * 3 * E, 4);          // qkv
  tot += bump_bytes((size_t)N * S * m->FF, 4);          // ffn hidden
  return tot + 256;
}

extern "C" int ff_encode(const ff_model* m, const float* input, const unsigned char* mask,
                         const int* kv_len, int N, int L, float* memory, void* workspace,
                         size_t workspace_bytes, ff_stream_t stream) {
  FF_RETURN_IF(check_model(m));
  FF_CHECK_ARG(N > 0 && L >= 0 && input && mask && memory && workspace, "ff_encode: bad arguments");
  const int E = m->E, FFd = m->FF, H = m->H, S = L + m->num_token;
  FF_CHECK_ARG(S <= m->pos_len, "ff_encode: S=%d exceeds the position table (%d rows)", S, m->pos_len);
  hipStream_t st = (hipStream_t)stream;
  FF_RETURN_IF(ff_gemm_prepare_stream(st));
  Bump bp(workspace, workspace_bytes);
  float* h1 = bp.take<float>((size_t)N * L * E);
  float* h2 = bp.take<float>((size_t)N * L * E);
  float* x = bp.take<float>((size_t)N * S * E);
  float* y = bp.take<float>((size_t)N * S * E);
  float* yq = bp.take<float>((size_t)N * S * E);
  float* o = bp.take<float>((size_t)N * S * E);
  float* qkv = bp.take<float>((size_t)N * S * 3 * E);
  float* hb = bp.take<float>((size_t)N * S * FFd);
  if (!bp.ok) { ff_set_error("ff_encode: workspace too small (%zu needed)", bp.off); return FF_ERR_WORKSPACE; }
  const int R = N * S;
  // a1: edge MLP (embedding.py:30-36) + token rows
  if (L > 0) {
    FF_RETURN_IF(gemm(input, m->in_dim, nullptr, 0, m->emb_w1, m->in_dim, m->emb_b1, nullptr, 0, h1, E, N * L, E,
                      m->in_dim, 1, st));
    FF_RETURN_IF(gemm(h1, E, nullptr, 0, m->emb_w2, E, m->emb_b2, nullptr, 0, h2, E, N * L, E, E, 0, st));
  }
  FF_RETURN_IF(ff_assemble_embedding(m->tok_embed, m->num_token, h2, E, N, L, E, x, st));
  // a4: pre-norm encoder layers (transformer.py:164-176)
  for (int l = 0; l < m->num_enc_layers; ++l) {
    const ff_layer_weights& w = m->enc[l];
    FF_RETURN_IF(ff_layernorm(x, E, w.norm1_w, w.norm1_b, m->ln_eps, y, E, yq, E, m->pos_table, E, 1, S, R, E, st));
    FF_RETURN_IF(gemm(yq, E, y, 2 * E, w.self_attn.in_proj_w, E, w.self_attn.in_proj_b, nullptr, 0, qkv, 3 * E, R,
                      3 * E, E, 0, st));
    ff_attn_desc d;
    memset(&d, 0, sizeof(d));
    d.q = qkv; d.k = qkv + E; d.v = qkv + 2 * E; d.o = o;
    d.ldq = d.ldk = d.ldv = 3 * E; d.ldo = E;
    d.num_groups = N; d.num_heads = H;
    d.nq = S; d.q_group_stride = S; d.q_inner = S; d.q_outer_stride = 0;
    d.nk = S; d.k_group_stride = S; d.k_stride = 1;
    d.kv_len = kv_len; d.key_mask = mask; d.mask_stride = S;
    d.scale = 0.125f;
    FF_RETURN_IF(ff_attention(&d, st));
    FF_RETURN_IF(gemm(o, E, nullptr, 0, w.self_attn.out_w, E, w.self_attn.out_b, x, E, x, E, R, E, E, 0, st));
    FF_RETURN_IF(ff_layernorm(x, E, w.norm2_w, w.norm2_b, m->ln_eps, y, E, nullptr, 0, nullptr, 0, 1, 1, R, E, st));
    FF_RETURN_IF(gemm(y, E, nullptr, 0, w.lin1_w, E, w.lin1_b, nullptr, 0, hb, FFd, R, FFd, E, 1, st));
    FF_RETURN_IF(gemm(hb, FFd, nullptr, 0, w.lin2_w, FFd, w.lin2_b, x, E, x, E, R, E, FFd, 0, st));
  }
  FF_RETURN_IF(ff_layernorm(x, E, m->enc_norm_w, m->enc_norm_b, m->ln_eps, memory, E, nullptr, 0, nullptr, 0, 1, 1,
                            R, E, st));
  return FF_OK;
}

extern "C" size_t ff_decode_workspace_bytes(const ff_model* m, const ff_decode_params* p, const int* num_input_host) {
  if (!m || !p || p->N <= 0 || p->F <= 0 || p->T <= 0) return 0;
  int btot = 0, max_bc = 0, nch = 0;
  plan_chunks(p, num_input_host, 1, nullptr, &btot, &max_bc, &nch);
  Bump bp(nullptr, 0);
  return layout_decode(m, p, engine_knobs(p), (size_t)btot, (size_t)max_bc, (size_t)nch, bp, nullptr) + 256;
}

extern "C" int ff_decode(const ff_model* m, const ff_decode_params* p, const float* memory,
                         const unsigned char* mask, const int* kv_len, const int* num_input,
                         const int* num_input_host, const unsigned char* extra_mask, int64_t* predict,
                         int* steps_done, int* step_counts, float* pointer_out, float* trace_logits,
                         float* trace_best, float* trace_second, int* seq_of_row, void* workspace,
                         size_t workspace_bytes, ff_stream_t stream) {
  FF_RETURN_IF(check_model(m));
  FF_CHECK_ARG(p != nullptr, "ff_decode: null params");
  FF_CHECK_ARG(p->variant == FF_PARALLEL || p->variant == FF_SEQ2SEQ, "ff_decode: bad variant");
  FF_CHECK_ARG(p->N > 0 && p->L >= 0 && p->F > 0 && p->T >= 1, "ff_decode: bad sizes");
  FF_CHECK_ARG(memory && mask && kv_len && predict && workspace, "ff_decode: null pointer");
  FF_CHECK_ARG(p->variant != FF_PARALLEL || num_input, "ff_decode: num_input required for the parallel variant");
  FF_CHECK_ARG(p->variant != FF_SEQ2SEQ || p->F == 1, "ff_decode: seq2seq decodes one sequence per wireframe");
  FF_CHECK_ARG(!p->stop_fn || (p->flags & FF_NO_STOP) || p->sync_every > 0, "ff_decode: stop_fn needs sync_every > 0");
  // The callback's cadence is a CONTRACT with callers that replay it elsewhere (an idle rank of a sharded decode joins the
  // same host collectives: faceformer_amd/dist.py check_points): the counters of the first n = enq - sync_every steps when
  // enq = 2 sync_every, 3 sync_every, ... steps are enqueued.  Both check paths below (host-mapped counters; drain + copy when
  // there are more counters than slots) keep that cadence for a stop_fn (tests: FF_PINNED_COUNTERS=8 in a child process).
  FF_CHECK_ARG(!(p->flags & FF_STOP_EACH_EOS) || p->variant == FF_SEQ2SEQ, "ff_decode: FF_STOP_EACH_EOS is a seq2seq rule");
  const int E = m->E, S = p->L + m->num_token, T = p->T, F = p->F, N = p->N;
  FF_CHECK_ARG(S <= m->pos_len, "ff_decode: S=%d exceeds the position table (%d rows)", S, m->pos_len);
  FF_CHECK_ARG(T - 1 <= m->qpos_len, "ff_decode: T-1=%d exceeds the query position table (%d rows)", T - 1, m->qpos_len);
  FF_CHECK_ARG(p->variant != FF_PARALLEL || F <= S, "ff_decode: F=%d anchors exceed S=%d", F, S);
  FF_CHECK_ARG(!(p->flags & FF_RETURN_POINTER) || pointer_out, "ff_decode: pointer_out required");
  // every padding-anchor sequence has its own row of an extra mask: no de-duplication then
  ff_decode_params prm_local = *p;
  if (extra_mask) prm_local.flags &= ~FF_DEDUP_PAD_ANCHORS;
  p = &prm_local;
  const bool dedup = p->variant == FF_PARALLEL && (p->flags & FF_DEDUP_PAD_ANCHORS) && num_input_host;
  hipStream_t main_st = (hipStream_t)stream;

  const int ns_req = plan_streams(p);
  std::vector<Chunk> chunks;
  int Btot = 0, max_bc = 0;
  plan_chunks(p, num_input_host, ns_req, &chunks, &Btot, &max_bc);
  Bump bp(workspace, workspace_bytes);
  DecodeBuffers buf;
  const int nch = (int)chunks.size();
  const EngineKnobs kn = engine_knobs(p);
  layout_decode(m, p, kn, (size_t)Btot, (size_t)max_bc, (size_t)nch, bp, &buf);
  if (!bp.ok) { ff_set_error("ff_decode: workspace too small (%zu needed, %zu given)", bp.off, workspace_bytes); return FF_ERR_WORKSPACE; }
  for (Chunk& c : chunks) {
    c.x0 = buf.x0_all + (size_t)T * c.b0 * E;
    c.qkv0 = buf.qkv0_all ? buf.qkv0_all + (size_t)T * c.b0 * 3 * E : nullptr;
    c.x0stat = buf.x0stat_all ? buf.x0stat_all + (size_t)c.b0 * (E / 32) * 2 : nullptr;
    const size_t ci = (size_t)(&c - chunks.data());
    const bool one = c.nw == 1 && buf.pg_all != nullptr;
    c.pg = one ? buf.pg_all + ci * (size_t)(p->L + m->num_token) * E : nullptr;
    c.pc = one ? buf.pc_all + ci * (size_t)((p->L + m->num_token + 3) & ~3) : nullptr;
  }
  const int ns = ns_req < (int)chunks.size() ? ns_req : (int)chunks.size();
  const bool forked = ns > 1;
  // With more than one stream ALL micro-batch work runs on the internal pool (the caller's stream is
  // often the legacy default stream, whose implicit synchronisation would serialise the others).
  hipStream_t sts[FF_MAX_STREAMS];
  sts[0] = main_st;
  StreamPool* pool = nullptr;
  FF_RETURN_IF(ff_gemm_prepare_stream(main_st));
  std::mutex* busy = pool_busy_mutex();
  FF_CHECK_ARG(busy, "ff_decode: no current device");
  std::lock_guard<std::mutex> one_decode_per_device(*busy);
  FF_RETURN_IF(pool_get(forked ? ns : 0, &pool));   // (also owns the pinned counter buffer / events of the stop check)
  if (forked)
    for (int s = 0; s < ns; ++s) sts[s] = pool->side[s];
  auto sync_all = [&]() -> int {
    for (int s = 0; s < ns; ++s) FF_CHECK_HIP(hipStreamSynchronize(sts[s]));
    return FF_OK;
  };

  int enq = 0;
  const bool each_eos = (p->flags & FF_STOP_EACH_EOS) != 0;
  // Everything that enqueues work on the side streams sits in this lambda: on ANY failure the streams are
  // drained before the error is returned (the caller frees the workspace the queued kernels use).
  auto run = [&]() -> int {
    // ---- per-batch invariants (main stream): memory + pos, cross-attention K|V of every layer ----
    const int RS = N * S;
    FF_RETURN_IF(ff_add_pos(memory, E, m->pos_table, E, 1, S, buf.mem_pos, E, RS, E, main_st));
    for (int l = 0; l < m->num_dec_layers; ++l) {
      const ff_mha_weights& c = m->dec[l].cross_attn;
      FF_RETURN_IF(gemm(buf.mem_pos, E, memory, E, c.in_proj_w + (size_t)E * E, E, c.in_proj_b + E, nullptr, 0,
                        buf.kvc[l], 2 * E, RS, 2 * E, E, 0, main_st));
      if (buf.kvp[l])
        FF_RETURN_IF(ff_attention_split_kv(buf.kvc[l], buf.kvc[l] + E, 2 * E, 2 * E, N, m->H, S, S, 1, buf.kvp[l], main_st));
    }
    // cnt_ge | cnt_eq | arrive | seen are consecutive in the workspace (layout_decode): ONE fill
    FF_CHECK_HIP(hipMemsetAsync(buf.cnt_ge, 0, (size_t)(reinterpret_cast<char*>(buf.seen + Btot) - reinterpret_cast<char*>(buf.cnt_ge)),
                                main_st));
    if (forked) {  // fork
      FF_CHECK_HIP(hipEventRecord(pool->fork_ev, main_st));
      for (int s = 0; s < ns; ++s) FF_CHECK_HIP(hipStreamWaitEvent(sts[s], pool->fork_ev, 0));
    }
    // pointer_fold operands of the one-wireframe micro-batches: G = memory_w W' ([S, E]; W' = the folded project weight, used
    // transposed), c = memory_w b'
    bool any_pg = false;
    for (const Chunk& c : chunks) any_pg = any_pg || c.pg != nullptr;
    if (any_pg) {
      FF_RETURN_IF(ff_transpose(m->proj_fold_w, E, E, E, buf.projT, E, main_st));
      if (forked) {
        FF_CHECK_HIP(hipEventRecord(pool->fork_ev, main_st));
        for (int s = 0; s < ns; ++s) FF_CHECK_HIP(hipStreamWaitEvent(sts[s], pool->fork_ev, 0));
      }
      for (const Chunk& c : chunks) {
        if (!c.pg) continue;
        const float* mem_w = memory + (size_t)c.w0 * S * E;
        FF_RETURN_IF(gemm(mem_w, E, nullptr, 0, buf.projT, E, nullptr, nullptr, 0, c.pg, E, S, E, E, 0, sts[c.sid]));
        FF_RETURN_IF(gemm(mem_w, E, nullptr, 0, m->proj_fold_b, E, nullptr, nullptr, 0, c.pc, 1, S, 1, E, 0, sts[c.sid]));
      }
    }
    // start tokens (anchors / SOS) and first decoder input rows of every micro-batch
    for (const Chunk& c : chunks) {
      hipLaunchKernelGGL(init_tokens_kernel, dim3(ff_cdiv(c.Bc, 256)), dim3(256), 0, sts[c.sid], buf.tok_all + c.b0,
                         c.Bc, c.Fc, c.f0, num_input ? num_input + c.w0 : nullptr, p->variant, m->num_token - 1,
                         p->tok_sos);
      FF_CHECK_LAUNCH();
      FF_RETURN_IF(ff_gather_rows(memory + (size_t)c.w0 * S * E, S, E, buf.tok_all + c.b0, c.Bc, c.Fc, c.x0, E,
                                  sts[c.sid]));
    }

    // ---- greedy loop -----------------------------------------------------------------------------------
    const int max_steps = T - 1;
    const bool dbg_timing = kn.dbg_timing;
    const auto host_t0 = std::chrono::steady_clock::now();
    // Stop rule on the host WITHOUT draining the queue and WITHOUT a copy launch: every pointer launch owns the counter of
    // its (step, micro-batch) and its last block stores the total into host-mapped pinned memory (ff_pointer_count_block).
    // Every sync_every steps an event is recorded behind the steps enqueued so far (on every stream); it is waited for
    // when another sync_every steps have been enqueued -- by then the host is a whole period ahead of it, so the wait
    // normally returns at once and the GPU always has a period of steps queued.  A stop is noticed at most
    // 2 * sync_every - 1 steps late; those surplus steps are dropped by the finalize kernels (exact results).
    bool stopped = false;
    int pending_enq = 0;   // > 0: events covering steps [0, pending_enq) are in flight
    const bool lagged = (size_t)T * (size_t)nch <= (size_t)kn.pinned;
    std::vector<int> tot;
    auto host_totals = [&](const int* per_chunk, int n) -> const int* {   // [n][nch] -> per-step totals
      tot.assign((size_t)n, 0);
      const volatile int* v = per_chunk;
      for (int s_ = 0; s_ < n; ++s_)
        for (int c = 0; c < nch; ++c) tot[(size_t)s_] += v[(size_t)s_ * nch + c];
      return tot.data();
    };
    auto eval_counts = [&](const int* cnt, int n) {
      if (p->stop_fn) return p->stop_fn(p->stop_user, cnt, n) != 0;   // the caller's (batch-global) rule
      if (p->variant == FF_PARALLEL) {
        for (int s = 0; s < n; ++s) if (cnt[s] == 0) return true;
      } else {
        int cum = 0;
        for (int s = 0; s < n; ++s) { cum += cnt[s]; if (cum == N) return true; }
      }
      return false;
    };
    auto enqueue_step = [&](int step) -> int {
      const int t = step + 1;
      for (const Chunk& c : chunks) {
        hipStream_t st = sts[c.sid];
        const Scratch& sc = buf.scr[c.sid];
        const size_t trow = (size_t)step * ((size_t)N * F) + c.b0;  // traces: step stride N*F (caller sizes them so)
        const size_t slot = (size_t)step * nch + (size_t)(&c - chunks.data());
        const bool folded_head = c.pg != nullptr && step_fuses(m, p, (long)t * c.Bc);
        float* logits_dst = trace_logits ? trace_logits + trow * S : sc.logits;
        ff_pointer_sync psync{each_eos ? buf.seen + c.b0 : nullptr, lagged ? buf.arrive + slot : nullptr,
                              lagged ? pool->hpin_dev + slot : nullptr, p->variant == FF_PARALLEL ? 0 : 1, c.x0stat,
                              folded_head ? 1 : 0};
        auto pointer_head = [&]() -> int {
          return ff_pointer_argmax_sync(
              folded_head ? nullptr : sc.p, E, memory + (size_t)c.w0 * S * E, S, E, mask + (size_t)c.w0 * S, kv_len + c.w0,
              extra_mask ? extra_mask + (size_t)c.b0 * S : nullptr, S, c.Bc, c.Fc,
              buf.tok_all + (size_t)t * Btot + c.b0, trace_best ? trace_best + trow : nullptr,
              trace_second ? trace_second + trow : nullptr, logits_dst, S,
              c.x0 + (size_t)t * c.Bc * E, E, buf.cnt_ge + slot, m->num_token, buf.cnt_eq + slot, p->tok_eos,
              (each_eos || lagged || c.x0stat || folded_head) ? &psync : nullptr, st);
        };
        FF_RETURN_IF(decoder_pass(m, p, kn, buf, sc, c, mask, kv_len, t, false, nullptr, st, folded_head ? logits_dst : nullptr));
        FF_RETURN_IF(pointer_head());
      }
      return FF_OK;
    };
    for (int step = 0; step < max_steps && !stopped;) {
      FF_RETURN_IF(enqueue_step(step));
      ++step;
      enq = step;
      if (p->sync_every > 0 && !(p->flags & FF_NO_STOP) && (enq % p->sync_every) == 0 && enq < max_steps) {
        if (lagged) {
          if (pending_enq > 0) {
            for (int s_ = 0; s_ < ns; ++s_) FF_CHECK_HIP(hipEventSynchronize(pool->chk_ev[s_]));
            stopped = eval_counts(host_totals(pool->hpin, pending_enq), pending_enq);
            pending_enq = 0;
          }
          if (!stopped) {
            for (int s_ = 0; s_ < ns; ++s_) FF_CHECK_HIP(hipEventRecord(pool->chk_ev[s_], sts[s_]));
            pending_enq = enq;
          }
        } else {   // more (step, micro-batch) counters than host slots: drain and copy
          // A caller's stop_fn is asked at the SAME cadence as on the lagged path (the first enq - sync_every steps, from
          // enq = 2 sync_every on): peers and idle ranks of a sharded decode replay exactly that sequence of host collectives
          // (dist.check_points).  The local rule has no such contract and looks at everything that has run.
          const int n_eval = p->stop_fn ? enq - p->sync_every : enq;
          if (n_eval > 0) {
            std::vector<int> hcnt((size_t)n_eval * nch);
            FF_RETURN_IF(sync_all());
            FF_CHECK_HIP(hipMemcpyAsync(hcnt.data(), (p->variant == FF_PARALLEL) ? buf.cnt_ge : buf.cnt_eq,
                                        sizeof(int) * hcnt.size(), hipMemcpyDeviceToHost, main_st));
            FF_CHECK_HIP(hipStreamSynchronize(main_st));
            stopped = eval_counts(host_totals(hcnt.data(), n_eval), n_eval);
          }
        }
      }
    }
    if (pending_enq > 0)   // the slots are reused by the next call
      for (int s_ = 0; s_ < ns; ++s_) FF_CHECK_HIP(hipEventSynchronize(pool->chk_ev[s_]));
    if (dbg_timing) {
      const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
      FF_RETURN_IF(sync_all());
      const double tot_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
      fprintf(stderr, "[ff_decode] host enqueue of %d steps x %zu chunks (%d of %d sequences decoded): %.2f ms; until GPU idle: "
                      "%.2f ms\n", enq, chunks.size(), Btot, N * F, host_ms, tot_ms);
    }
    if (forked) {  // join
      for (int s = 0; s < ns; ++s) {
        FF_CHECK_HIP(hipEventRecord(pool->join_ev[s], sts[s]));
        FF_CHECK_HIP(hipStreamWaitEvent(main_st, pool->join_ev[s], 0));
      }
    }
    return FF_OK;
  };
  {
    const int rc = run();
    if (rc != FF_OK) {
      for (int s = 0; s < ns; ++s) (void)hipStreamSynchronize(sts[s]);
      (void)hipStreamSynchronize(main_st);
      return rc;
    }
  }

  // Everything after the greedy loop (stop step, packing, the optional return-pointer pass) enqueues work that reads the
  // caller's workspace as well: same rule as above -- on any failure the streams are drained before the error goes back.
  auto finish = [&]() -> int {
    hipLaunchKernelGGL(steps_kernel, dim3(1), dim3(64), 0, main_st, buf.cnt_ge, buf.cnt_eq, nch, p->variant, N, enq,
                       ((p->flags & FF_NO_STOP) || p->stop_fn) ? 1 : 0, buf.cnt_tot, buf.steps_dev);
    FF_CHECK_LAUNCH();
    for (const Chunk& c : chunks) {
      const long total = (long)c.nw * F * T;
      const int grid = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
      hipLaunchKernelGGL(finalize_chunk_kernel, dim3(grid), dim3(256), 0, main_st, buf.tok_all, Btot, T, buf.steps_dev,
                         num_input, dedup ? 1 : 0, F, c.w0, c.nw, c.Fc, c.f0, c.b0, predict, seq_of_row);
      FF_CHECK_LAUNCH();
    }
    int steps = 0;
    FF_CHECK_HIP(hipMemcpyAsync(&steps, buf.steps_dev, sizeof(int), hipMemcpyDeviceToHost, main_st));
    if (step_counts && enq > 0)
      FF_CHECK_HIP(hipMemcpyAsync(step_counts, buf.cnt_tot, sizeof(int) * enq, hipMemcpyDeviceToHost, main_st));
    FF_CHECK_HIP(hipStreamSynchronize(main_st));
    if (steps_done) *steps_done = steps;

    // ---- optional: project(decoder(...)) of every prefix row at the last executed step
    //      (SurfaceFormer returns it as inputs['pointer'], reference model.py:217) --------------------
    if ((p->flags & FF_RETURN_POINTER) && steps > 0) {
      FF_CHECK_ARG(m->FF >= m->E, "ff_decode: FF_RETURN_POINTER needs FF >= E");
      FF_CHECK_ARG(Btot == N * F, "ff_decode: FF_RETURN_POINTER is not available with de-duplicated sequences");
      for (const Chunk& c : chunks) {
        const Scratch& sc = buf.scr[0];
        // one micro-batch: its [steps * Bc, E] rows ARE pointer_out [steps, Btot, E]; several: through the FF-wide scratch
        // and one strided copy per micro-batch (was one copy launch per position: 258 of them for configs A / D)
        float* proj_all = nch == 1 ? pointer_out : sc.h;
        FF_RETURN_IF(decoder_pass(m, p, kn, buf, sc, c, mask, kv_len, steps, true, proj_all, main_st));
        if (nch > 1)
          FF_CHECK_HIP(hipMemcpy2DAsync(pointer_out + (size_t)c.b0 * E, sizeof(float) * (size_t)Btot * E, proj_all,
                                        sizeof(float) * (size_t)c.Bc * E, sizeof(float) * (size_t)c.Bc * E, (size_t)steps,
                                        hipMemcpyDeviceToDevice, main_st));
      }
    }
    return FF_OK;
  };
  {
    const int rc = finish();
    if (rc != FF_OK) {
      for (int s = 0; s < ns; ++s) (void)hipStreamSynchronize(sts[s]);
      (void)hipStreamSynchronize(main_st);
      return rc;
    }
  }
  return FF_OK;
}
