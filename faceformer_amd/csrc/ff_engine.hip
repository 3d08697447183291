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
#include <chrono>
#include <cstdlib>
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
__global__ void init_tokens_kernel(int* tok, int B, int F, const int* num_input, int variant,
                                   int pad_tok, int sos, int b_off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  if (variant == FF_PARALLEL) {
    // anchors = arange(F) per wireframe, WITHOUT the +num_token offset (reference quirk C-3,
    // model_para.py:201); rows >= num_input[w] start from token num_token-1 (model_para.py:204-205)
    const int b = b_off + i;
    const int w = b / F, f = b % F;
    tok[i] = f < num_input[w] ? f : pad_tok;
  } else {
    tok[i] = sos;
  }
}

// steps_done from the per-step counters, then predict[b, j] (int64) = token or 0 after the stop.
__global__ void finalize_kernel(const int* __restrict__ tok_all, const int* __restrict__ cnt_ge,
                                const int* __restrict__ cnt_eq, int variant, int N, int Btot, int T,
                                int steps_enqueued, int no_stop, int64_t* __restrict__ predict,
                                int* __restrict__ steps_done_out) {
  int steps = steps_enqueued;
  if (no_stop) {
  } else if (variant == FF_PARALLEL) {
    for (int s = 0; s < steps_enqueued; ++s)
      if (cnt_ge[s] == 0) { steps = s + 1; break; }
  } else {
    int cum = 0;
    for (int s = 0; s < steps_enqueued; ++s) {
      cum += cnt_eq[s];
      if (cum == N) { steps = s + 1; break; }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *steps_done_out = steps;
  const size_t total = (size_t)Btot * T;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / T), j = (int)(i % T);
    predict[i] = (j <= steps) ? (int64_t)tok_all[(size_t)j * Btot + b] : (int64_t)0;
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
  return FF_OK;
}

struct Linear { const float* w; const float* b; };

int gemm(const float* A, int lda, const float* A2, int n_split, const float* W, int ldw,
         const float* bias, const float* res, int ldr, float* C, int ldc, int M, int N, int K, int act,
         hipStream_t st) {
  return ff_gemm_f32(A, lda, A2, n_split, W, ldw, bias, res, ldr, C, ldc, M, N, K, act, 0, st);
}

// The same product through the 3 x bf16 kernel when the weight's planes are bound and the launch is in the
// range where it wins (measured on MI355X: N or K >= 1024 from ~4096 rows on; tools/bench_gemm.py).
int gemm_or_x3(const ff_decode_params* prm, const void* planes, const float* A, int lda, const float* A2, int n_split,
               const float* W, int ldw, const float* bias, const float* res, int ldr, float* C, int ldc, int M,
               int N, int K, int act, hipStream_t st) {
  // x3_min_rows is the threshold of the widest product (N >= 1536: 112 vs 104 TF/s at 4096 rows); the K = 1024
  // product needs 1.5x, the N = 1024 one 2x and the N = K = 512 ones 4x as many rows before the larger tiles pay
  const long need = (long)prm->x3_min_rows * (N >= 1536 ? 2 : (K >= 1024 ? 3 : (N >= 1024 ? 4 : 8))) / 2;
  if (planes && prm->x3_min_rows > 0 && M >= need && (K % 32) == 0 && K >= 64 && (!A2 || (n_split % 128) == 0))
    return ff_gemm_x3(A, lda, A2, n_split, planes, bias, res, ldr, C, ldc, M, N, K, act, st);
  return gemm(A, lda, A2, n_split, W, ldw, bias, res, ldr, C, ldc, M, N, K, act, st);
}

// Scratch of one in-flight micro-batch (one set per stream): R = t*Bc active rows, position-major.
struct Scratch {
  float *x, *y, *yq, *qkv, *o, *h, *p, *logits;
};

struct DecodeBuffers {
  float *mem_pos, *kvc[FF_MAX_LAYERS];
  float *x0_all, *qkv0_all;
  int* tok_all;   // [T, Btot] global, position-major
  Scratch scr[FF_MAX_STREAMS];
  int *cnt_ge, *cnt_eq, *steps_dev;
};

// A micro-batch is a contiguous range [b0, b0 + Bc) of the global sequence index b = w*F + f:
// either whole wireframes (nw >= 1, Fc = F) or a group of Fc < F sequences of ONE wireframe.
struct Chunk {
  int w0, nw, Fc, b0, Bc, sid;
  float* x0;     // [T, Bc, E]
  float* qkv0;   // [T, Bc, 3E] or null
};

struct Plan {
  int cw, cs, ns;       // wireframes per chunk, sequences per sub-wireframe chunk (0: none), streams
  size_t max_bc;
};

Plan make_plan(const ff_decode_params* p) {
  Plan pl;
  pl.cw = (p->chunk_wireframes <= 0 || p->chunk_wireframes > p->N) ? p->N : p->chunk_wireframes;
  pl.cs = (p->chunk_seqs > 0 && p->chunk_seqs < p->F) ? p->chunk_seqs : 0;
  if (pl.cs) pl.cw = 1;
  pl.ns = p->num_streams < 1 ? 1 : (p->num_streams > FF_MAX_STREAMS ? FF_MAX_STREAMS : p->num_streams);
  pl.max_bc = pl.cs ? (size_t)pl.cs : (size_t)pl.cw * p->F;
  return pl;
}

size_t layout_decode(const ff_model* m, const ff_decode_params* p, Bump& bp, DecodeBuffers* out) {
  const int E = m->E, FFd = m->FF, S = p->L + m->num_token, T = p->T;
  const size_t Btot = (size_t)p->N * p->F;
  const Plan pl = make_plan(p);
  const size_t Bch = pl.max_bc;
  const size_t Rmax = (size_t)(T - 1 > 0 ? T - 1 : 1) * Bch;
  DecodeBuffers b;
  memset(&b, 0, sizeof(b));
  b.mem_pos = bp.take<float>((size_t)p->N * S * E);
  for (int l = 0; l < m->num_dec_layers; ++l) b.kvc[l] = bp.take<float>((size_t)p->N * S * 2 * E);
  b.x0_all = bp.take<float>((size_t)T * Btot * E);
  b.tok_all = bp.take<int>((size_t)T * Btot);
  b.qkv0_all = (p->flags & FF_REUSE_LAYER0_QKV) ? bp.take<float>((size_t)T * Btot * 3 * E) : nullptr;
  for (int s = 0; s < pl.ns; ++s) {
    Scratch& c = b.scr[s];
    c.x = bp.take<float>(Rmax * E);
    c.y = bp.take<float>(Rmax * E);
    c.yq = bp.take<float>(Rmax * E);
    c.qkv = bp.take<float>(Rmax * 3 * E);
    c.o = bp.take<float>(Rmax * E);
    c.h = bp.take<float>(Rmax * FFd);
    c.p = bp.take<float>(Bch * E);
    c.logits = bp.take<float>(Bch * (size_t)S);
  }
  b.cnt_ge = bp.take<int>(T);
  b.cnt_eq = bp.take<int>(T);
  b.steps_dev = bp.take<int>(4);
  if (out) *out = b;
  return bp.off;
}

// One decoder pass over the current prefix (t positions) of one micro-batch.
// full_rows: evaluate every layer for all rows and project all rows into proj_all (ld = E rows
// position-major within the chunk); otherwise the result is p[Bc, E] for the newest position.
int decoder_pass(const ff_model* m, const ff_decode_params* prm, const DecodeBuffers& bufs, const Scratch& buf,
                 const Chunk& ck, const unsigned char* mask, const int* kv_len, int t, bool full_rows,
                 float* proj_all, hipStream_t st) {
  const int E = m->E, FFd = m->FF, H = m->H, S = prm->L + m->num_token, F = ck.Fc, T = prm->T;
  const int Bc = ck.Bc, R = t * Bc, nd = m->num_dec_layers;
  const size_t newoff = (size_t)(t - 1) * Bc;
  const bool reuse0 = (prm->flags & FF_REUSE_LAYER0_QKV) != 0 && ck.qkv0 != nullptr;
  const bool prune_last = (prm->flags & FF_LAST_LAYER_LAST_ROW) != 0 && !full_rows;
  const float* qpos = m->qpos_table;
  const float* qpos_new = qpos + (size_t)(t - 1) * E;

  for (int l = 0; l < nd; ++l) {
    const ff_layer_weights& w = m->dec[l];
    const bool last = prune_last && (l == nd - 1);
    const float* xin = (l == 0) ? ck.x0 : buf.x;
    const float* QKV;
    // ---- self attention: q = k = LN1(x) + qpos, v = LN1(x), no mask (transformer.py:242-246) ----
    if (l == 0 && reuse0) {
      FF_RETURN_IF(ff_layernorm(xin + newoff * E, E, w.norm1_w, w.norm1_b, m->ln_eps, buf.y, E, buf.yq, E,
                                qpos_new, E, Bc, 1, Bc, E, st));
      FF_RETURN_IF(gemm(buf.yq, E, buf.y, 2 * E, w.self_attn.in_proj_w, E, w.self_attn.in_proj_b, nullptr, 0,
                        ck.qkv0 + newoff * 3 * E, 3 * E, Bc, 3 * E, E, 0, st));
      QKV = ck.qkv0;
    } else {
      FF_RETURN_IF(ff_layernorm(xin, E, w.norm1_w, w.norm1_b, m->ln_eps, buf.y, E, buf.yq, E, qpos, E, Bc, T,
                                R, E, st));
      FF_RETURN_IF(gemm_or_x3(prm, w.in_proj_planes, buf.yq, E, buf.y, 2 * E, w.self_attn.in_proj_w, E,
                              w.self_attn.in_proj_b, nullptr, 0, buf.qkv, 3 * E, R, 3 * E, E, 0, st));
      QKV = buf.qkv;
    }
    // rows that continue through the rest of this layer
    const size_t roff = last ? newoff : 0;
    const int Rl = last ? Bc : R;
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
    FF_RETURN_IF(gemm_or_x3(prm, w.self_out_planes, buf.o + roff * E, E, nullptr, 0, w.self_attn.out_w, E,
                            w.self_attn.out_b, xin + roff * E, E, buf.x + roff * E, E, Rl, E, E, 0, st));
    // ---- cross attention: q = LN2(x) + qpos, k = memory + pos, v = memory (transformer.py:247-252);
    //      K/V come from the per-batch cache ----
    if (last)
      FF_RETURN_IF(ff_layernorm(buf.x + roff * E, E, w.norm2_w, w.norm2_b, m->ln_eps, nullptr, 0, buf.yq + roff * E,
                                E, qpos_new, E, Bc, 1, Rl, E, st));
    else
      FF_RETURN_IF(ff_layernorm(buf.x, E, w.norm2_w, w.norm2_b, m->ln_eps, nullptr, 0, buf.yq, E, qpos, E, Bc, T,
                                Rl, E, st));
    float* qc = buf.qkv;  // [rows, E] view of the scratch
    FF_RETURN_IF(gemm_or_x3(prm, w.cross_q_planes, buf.yq + roff * E, E, nullptr, 0, w.cross_attn.in_proj_w, E,
                            w.cross_attn.in_proj_b, nullptr, 0, qc + roff * E, E, Rl, E, E, 0, st));
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
      FF_RETURN_IF(ff_attention(&d, st));
    }
    FF_RETURN_IF(gemm_or_x3(prm, w.cross_out_planes, buf.o + roff * E, E, nullptr, 0, w.cross_attn.out_w, E,
                            w.cross_attn.out_b, buf.x + roff * E, E, buf.x + roff * E, E, Rl, E, E, 0, st));
    // ---- feed forward (transformer.py:253-255) ----
    FF_RETURN_IF(ff_layernorm(buf.x + roff * E, E, w.norm3_w, w.norm3_b, m->ln_eps, buf.y + roff * E, E, nullptr, 0,
                              nullptr, 0, 1, 1, Rl, E, st));
    FF_RETURN_IF(gemm_or_x3(prm, w.lin1_planes, buf.y + roff * E, E, nullptr, 0, w.lin1_w, E, w.lin1_b, nullptr, 0,
                            buf.h + roff * FFd, FFd, Rl, FFd, E, 1, st));
    FF_RETURN_IF(gemm_or_x3(prm, w.lin2_planes, buf.h + roff * FFd, FFd, nullptr, 0, w.lin2_w, FFd, w.lin2_b,
                            buf.x + roff * E, E, buf.x + roff * E, E, Rl, E, FFd, 0, st));
  }
  // ---- decoder.norm + project (transformer.py:115-116, model_para.py:225) ----
  if (full_rows) {
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

// Internal side streams + fork/join events (created once per process).
struct StreamPool {
  hipStream_t side[FF_MAX_STREAMS];
  hipEvent_t fork_ev, join_ev[FF_MAX_STREAMS];
  int created;
  bool events;
};
StreamPool g_pool = {{}, nullptr, {}, 0, false};

int pool_get(int n) {
  if (!g_pool.events) {
    FF_CHECK_HIP(hipEventCreateWithFlags(&g_pool.fork_ev, hipEventDisableTiming));
    for (int i = 0; i < FF_MAX_STREAMS; ++i)
      FF_CHECK_HIP(hipEventCreateWithFlags(&g_pool.join_ev[i], hipEventDisableTiming));
    g_pool.events = true;
  }
  while (g_pool.created < n) {
    FF_CHECK_HIP(hipStreamCreateWithFlags(&g_pool.side[g_pool.created], hipStreamNonBlocking));
    g_pool.created++;
  }
  return FF_OK;
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

extern "C" size_t ff_decode_workspace_bytes(const ff_model* m, const ff_decode_params* p) {
  if (!m || !p || p->N <= 0 || p->F <= 0 || p->T <= 0) return 0;
  Bump bp(nullptr, 0);
  return layout_decode(m, p, bp, nullptr) + 256;
}

extern "C" int ff_decode(const ff_model* m, const ff_decode_params* p, const float* memory,
                         const unsigned char* mask, const int* kv_len, const int* num_input,
                         const unsigned char* extra_mask, int64_t* predict, int* steps_done,
                         int* step_counts, float* pointer_out, float* trace_logits, float* trace_best,
                         float* trace_second, void* workspace, size_t workspace_bytes,
                         ff_stream_t stream) {
  FF_RETURN_IF(check_model(m));
  FF_CHECK_ARG(p != nullptr, "ff_decode: null params");
  FF_CHECK_ARG(p->variant == FF_PARALLEL || p->variant == FF_SEQ2SEQ, "ff_decode: bad variant");
  FF_CHECK_ARG(p->N > 0 && p->L >= 0 && p->F > 0 && p->T >= 1, "ff_decode: bad sizes");
  FF_CHECK_ARG(memory && mask && kv_len && predict && workspace, "ff_decode: null pointer");
  FF_CHECK_ARG(p->variant != FF_PARALLEL || num_input, "ff_decode: num_input required for the parallel variant");
  FF_CHECK_ARG(p->variant != FF_SEQ2SEQ || p->F == 1, "ff_decode: seq2seq decodes one sequence per wireframe");
  const int E = m->E, S = p->L + m->num_token, T = p->T, F = p->F, N = p->N;
  FF_CHECK_ARG(S <= m->pos_len, "ff_decode: S=%d exceeds the position table (%d rows)", S, m->pos_len);
  FF_CHECK_ARG(T - 1 <= m->qpos_len, "ff_decode: T-1=%d exceeds the query position table (%d rows)", T - 1, m->qpos_len);
  FF_CHECK_ARG(p->variant != FF_PARALLEL || F <= S, "ff_decode: F=%d anchors exceed S=%d", F, S);
  FF_CHECK_ARG(!(p->flags & FF_RETURN_POINTER) || pointer_out, "ff_decode: pointer_out required");
  hipStream_t main_st = (hipStream_t)stream;

  Bump bp(workspace, workspace_bytes);
  DecodeBuffers buf;
  layout_decode(m, p, bp, &buf);
  if (!bp.ok) { ff_set_error("ff_decode: workspace too small (%zu needed, %zu given)", bp.off, workspace_bytes); return FF_ERR_WORKSPACE; }

  const int Btot = N * F;
  const Plan pl = make_plan(p);
  std::vector<Chunk> chunks;
  for (int w0 = 0; w0 < N; w0 += pl.cw) {
    const int nw = (N - w0) < pl.cw ? (N - w0) : pl.cw;
    const int fstep = pl.cs ? pl.cs : F;
    for (int f0 = 0; f0 < F; f0 += fstep) {
      Chunk c;
      c.w0 = w0; c.nw = nw;
      c.Fc = (F - f0) < fstep ? (F - f0) : fstep;
      c.b0 = w0 * F + f0;
      c.Bc = nw * c.Fc;
      c.sid = (int)(chunks.size() % pl.ns);
      c.x0 = buf.x0_all + (size_t)T * c.b0 * E;
      c.qkv0 = buf.qkv0_all ? buf.qkv0_all + (size_t)T * c.b0 * 3 * E : nullptr;
      chunks.push_back(c);
    }
  }
  const int ns = pl.ns < (int)chunks.size() ? pl.ns : (int)chunks.size();
  // With more than one stream ALL micro-batch work runs on the internal pool (the caller's stream is
  // often the legacy default stream, whose implicit synchronisation would serialise the others).
  hipStream_t sts[FF_MAX_STREAMS];
  sts[0] = main_st;
  if (ns > 1) {
    FF_RETURN_IF(pool_get(ns));
    for (int s = 0; s < ns; ++s) sts[s] = g_pool.side[s];
  }

  // ---- per-batch invariants (main stream): memory + pos, cross-attention K|V of every layer ----
  const int RS = N * S;
  FF_RETURN_IF(ff_add_pos(memory, E, m->pos_table, E, 1, S, buf.mem_pos, E, RS, E, main_st));
  for (int l = 0; l < m->num_dec_layers; ++l) {
    const ff_mha_weights& c = m->dec[l].cross_attn;
    FF_RETURN_IF(gemm(buf.mem_pos, E, memory, E, c.in_proj_w + (size_t)E * E, E, c.in_proj_b + E, nullptr, 0,
                      buf.kvc[l], 2 * E, RS, 2 * E, E, 0, main_st));
  }
  FF_CHECK_HIP(hipMemsetAsync(buf.cnt_ge, 0, sizeof(int) * T, main_st));
  FF_CHECK_HIP(hipMemsetAsync(buf.cnt_eq, 0, sizeof(int) * T, main_st));
  // start tokens (anchors / SOS) for every sequence
  hipLaunchKernelGGL(init_tokens_kernel, dim3(ff_cdiv(Btot, 256)), dim3(256), 0, main_st, buf.tok_all, Btot, F,
                     num_input, p->variant, m->num_token - 1, p->tok_sos, 0);
  FF_CHECK_LAUNCH();
  if (ns > 1) {  // fork
    FF_CHECK_HIP(hipEventRecord(g_pool.fork_ev, main_st));
    for (int s = 0; s < ns; ++s) FF_CHECK_HIP(hipStreamWaitEvent(sts[s], g_pool.fork_ev, 0));
  }
  // first decoder input rows
  for (const Chunk& c : chunks)
    FF_RETURN_IF(ff_gather_rows(memory + (size_t)c.w0 * S * E, S, E, buf.tok_all + c.b0, c.Bc, c.Fc, c.x0, E,
                                sts[c.sid]));

  auto sync_all = [&]() -> int {
    for (int s = 0; s < ns; ++s) FF_CHECK_HIP(hipStreamSynchronize(sts[s]));
    return FF_OK;
  };

  // ---- greedy loop -----------------------------------------------------------------------------------
  const int max_steps = T - 1;
  const bool dbg_timing = getenv("FF_DEBUG_TIMING") != nullptr;
  const auto host_t0 = std::chrono::steady_clock::now();
  int enq = 0;
  std::vector<int> hcnt(T > 0 ? T : 1);
  bool stopped = false;
  for (int step = 0; step < max_steps && !stopped; ++step) {
    const int t = step + 1;
    for (const Chunk& c : chunks) {
      hipStream_t st = sts[c.sid];
      const Scratch& sc = buf.scr[c.sid];
      FF_RETURN_IF(decoder_pass(m, p, buf, sc, c, mask, kv_len, t, false, nullptr, st));
      const size_t trow = (size_t)step * Btot + c.b0;
      FF_RETURN_IF(ff_pointer_argmax(
          sc.p, E, memory + (size_t)c.w0 * S * E, S, E, mask + (size_t)c.w0 * S, kv_len + c.w0,
          extra_mask ? extra_mask + (size_t)c.b0 * S : nullptr, S, c.Bc, c.Fc,
          buf.tok_all + (size_t)t * Btot + c.b0, trace_best ? trace_best + trow : nullptr,
          trace_second ? trace_second + trow : nullptr, trace_logits ? trace_logits + trow * S : sc.logits, S,
          c.x0 + (size_t)t * c.Bc * E, E, buf.cnt_ge + step, m->num_token, buf.cnt_eq + step, p->tok_eos, st));
    }
    enq = step + 1;
    if (p->sync_every > 0 && !(p->flags & FF_NO_STOP) && (enq % p->sync_every) == 0 && enq < max_steps) {
      FF_RETURN_IF(sync_all());
      const int* src = (p->variant == FF_PARALLEL) ? buf.cnt_ge : buf.cnt_eq;
      FF_CHECK_HIP(hipMemcpyAsync(hcnt.data(), src, sizeof(int) * enq, hipMemcpyDeviceToHost, main_st));
      FF_CHECK_HIP(hipStreamSynchronize(main_st));
      if (p->variant == FF_PARALLEL) {
        for (int s = 0; s < enq; ++s) if (hcnt[s] == 0) { stopped = true; break; }
      } else {
        int cum = 0;
        for (int s = 0; s < enq; ++s) { cum += hcnt[s]; if (cum == N) { stopped = true; break; } }
      }
    }
  }
  if (dbg_timing) {
    const double host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    FF_CHECK_HIP(hipStreamSynchronize(main_st));
    const double tot_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
    fprintf(stderr, "[ff_decode] host enqueue of %d steps x %zu chunks: %.2f ms; until GPU idle: %.2f ms\n", enq,
            chunks.size(), host_ms, tot_ms);
  }
  if (ns > 1) {  // join
    for (int s = 0; s < ns; ++s) {
      FF_CHECK_HIP(hipEventRecord(g_pool.join_ev[s], sts[s]));
      FF_CHECK_HIP(hipStreamWaitEvent(main_st, g_pool.join_ev[s], 0));
    }
  }

  hipLaunchKernelGGL(finalize_kernel, dim3(ff_cdiv(Btot * T, 256) < 1024 ? ff_cdiv(Btot * T, 256) : 1024),
                     dim3(256), 0, main_st, buf.tok_all, buf.cnt_ge, buf.cnt_eq, p->variant, N, Btot, T, enq,
                     (p->flags & FF_NO_STOP) ? 1 : 0, predict, buf.steps_dev);
  FF_CHECK_LAUNCH();
  int steps = 0;
  FF_CHECK_HIP(hipMemcpyAsync(&steps, buf.steps_dev, sizeof(int), hipMemcpyDeviceToHost, main_st));
  if (step_counts && enq > 0)
    FF_CHECK_HIP(hipMemcpyAsync(step_counts, (p->variant == FF_PARALLEL) ? buf.cnt_ge : buf.cnt_eq,
                                sizeof(int) * enq, hipMemcpyDeviceToHost, main_st));
  FF_CHECK_HIP(hipStreamSynchronize(main_st));
  if (steps_done) *steps_done = steps;

  // ---- optional: project(decoder(...)) of every prefix row at the last executed step
  //      (SurfaceFormer returns it as inputs['pointer'], reference model.py:217) --------------------
  if ((p->flags & FF_RETURN_POINTER) && steps > 0) {
    FF_CHECK_ARG(m->FF >= m->E, "ff_decode: FF_RETURN_POINTER needs FF >= E");
    for (const Chunk& c : chunks) {
      const Scratch& sc = buf.scr[0];
      float* proj_all = sc.h;  // [steps*Bc, E] fits in the FF-wide scratch
      FF_RETURN_IF(decoder_pass(m, p, buf, sc, c, mask, kv_len, steps, true, proj_all, main_st));
      for (int j = 0; j < steps; ++j)
        FF_CHECK_HIP(hipMemcpyAsync(pointer_out + ((size_t)j * Btot + c.b0) * E, proj_all + (size_t)j * c.Bc * E,
                                    sizeof(float) * c.Bc * E, hipMemcpyDeviceToDevice, main_st));
    }
  }
  return FF_OK;
}
