/*
 * faceformer_hip.h -- C ABI of libfaceformer_hip.so (gfx950 / MI355X).
 *
 * The reference (manycore-research/faceformer) is pure Python on torch and has NO FFI: the
 * boundary of its decode path is the Python module surface `model_class(**cfg.model)(batch)`
 * (reference faceformer/trainer.py:20,27-28).  This header is the native layer underneath the
 * drop-in Python modules of `faceformer_amd`: every entry point replaces a group of torch operator
 * call sites of the reference (cited per function), takes plain device pointers / sizes and a
 * hipStream_t, returns 0 on success (negative ff_status otherwise) and never synchronises the device
 * -- except ff_decode(), which owns the greedy loop and may wait on its own stream every `sync_every`
 * steps to evaluate the reference's stop rule.  Device memory comes from the caller (workspaces); the
 * one exception is the partial-tile exchange buffer of the stream-K projection kernels, allocated once
 * per (device, stream) at the first launch on that stream.
 *
 * All tensors are fp32 row-major unless stated; "ld*" are leading dimensions in ELEMENTS.
 * Pointers must be 16-byte aligned and every ld / K / E a multiple of 4 (float4 access).
 */
#ifndef FACEFORMER_HIP_H
#define FACEFORMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ff_stream_t; /* hipStream_t */

enum ff_status {
  FF_OK = 0,
  FF_ERR_ARG = -1,       /* bad size / alignment / unsupported shape */
  FF_ERR_LAUNCH = -2,    /* hip launch or runtime error (see ff_last_error) */
  FF_ERR_WORKSPACE = -3, /* workspace too small */
  FF_ERR_NO_DEVICE = -4
};

#define FF_HEAD_DIM 64   /* attention head width of the MFMA kernels (reference: 512 / 8); other widths: ff_attention_general */
#define FF_MAX_LAYERS 16

/* Library version (major*10000 + minor*100 + patch).  103: ff_attention_general / ff_attn_general_desc added (round 6).  101: the struct layouts of this header (round 5: ff_decode_params lost
 * chain_max_rows / flow_min_rows, FF_STOP_EACH_EOS added; round 4: ff_layer_weights grew by the ln*_planes / ln*_csum
 * pointers).  A caller built against another header must refuse to run: hip/lib.py asserts equality with FF_ABI_VERSION. */
#define FF_ABI_VERSION 103
int ff_version(void);
/* Thread-local text of the last error returned by this library ("" if none). */
const char* ff_last_error(void);
/* Number of visible HIP devices (0 when none; never fails). */
int ff_device_count(void);
/* Measurement hooks (bench.py roofline leg; not on the product path).  Between begin and end every
 * op launch of this library is bracketed by a hipEvent pair on its stream; end() synchronises the
 * device and returns, per category (0 f32 gemm, 1 attention, 2 layernorm, 3 pointer, 4 other row ops, 5 unused (the chain
 * launches of round 3, removed), 6 the 3 x bf16 split gemm), the summed kernel time [ms], algorithmic work (flops for 0/1/3/6, bytes for 2/4) and launches. */
int ff_profile_begin(void);
int ff_profile_end(double* ms_by_cat, double* work_by_cat, long long* launches_by_cat, int ncat);
/* Algorithmic bytes (operands read once + results written once) summed per category since the last
 * ff_profile_begin (only the two GEMM categories are filled in). */
int ff_profile_bytes(double* bytes_by_cat, int ncat);
/* Mean event-pair interval [us] around an EMPTY kernel, `launches` of them queued back to back on `stream`: what the
 * event bracket adds per launch to the category times of ff_profile_end (bench.py reports times net of it). */
int ff_profile_bracket_us(int launches, double* us_per_launch, ff_stream_t stream);
/* Effective shader clock UNDER LOAD (measurement hook): launch() enqueues a one-wave kernel on `stream` -- a stream of its own,
 * beside the kernels under test -- that reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
 * (s_memrealtime), sleeps `spin_us` and reads both again; read() waits for `stream` and returns shader cycles / wall time in
 * GHz (and the measured interval).  The chip clocks to its power budget: this is the clock a roofline has to be priced at. */
int ff_clock_probe_launch(double spin_us, ff_stream_t stream);
int ff_clock_probe_read(double* ghz, double* measured_us, ff_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * G2  LayerNorm (+ positional add).  Replaces nn.LayerNorm followed by `with_pos_embed`
 * (reference faceformer/transformer.py:168-169, 242-243, 247-249, 253; torch LayerNorm: eps inside
 * the sqrt, biased variance, affine).
 *   y[r,:]    = LN(x[r,:]) * gamma + beta                      (skipped if y    == NULL)
 *   ypos[r,:] = y[r,:] + pos[((r / pos_div) % pos_mod), :]     (skipped if ypos == NULL)
 * One wavefront per row, float4 loads, shuffle reductions.  E % 4 == 0, E <= 2048.
 * ------------------------------------------------------------------------------------------- */
int ff_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float eps,
                 float* y, int ldy, float* ypos, int ldypos,
                 const float* pos, int ldpos, int pos_div, int pos_mod,
                 int rows, int E, ff_stream_t stream);

/* out[r,:] = x[r,:] + pos[((r / pos_div) % pos_mod), :]   (`memory + pos`, transformer.py:249) */
int ff_add_pos(const float* x, int ldx, const float* pos, int ldpos, int pos_div, int pos_mod,
               float* out, int ldout, int rows, int E, ff_stream_t stream);

/* x[r, 0:E] = gelu(x[r, 0:E]) in place, exact erf form: 0.5 x (1 + erf(x / sqrt 2)) (torch F.gelu's default; the "gelu"
 * activation of reference transformer.py:276-284 -- module surface only, no reference config uses it).  E % 4 == 0. */
int ff_gelu(float* x, int ldx, int rows, int E, ff_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * G3  Dense projection on the f32 matrix cores (v_mfma_f32_32x32x2_f32).  Replaces F.linear /
 * addmm call sites (in-proj q,k,v, out-proj, linear1+relu, linear2, project, embedding MLP;
 * reference transformer.py:170-175, 244-255, embedding.py:30-36, model_para.py:225):
 *   C[m,n] = act( sum_k Asel[m,k] * W[n,k] + bias[n] ) + residual[m,n]
 * W is the nn.Linear weight layout [N,K]; Asel = A for n < n_split and A2 for n >= n_split
 * (lets one launch produce q,k from `LN(x)+pos` and v from `LN(x)`); pass A2 = NULL to disable.
 * bias / residual may be NULL.  residual may alias C.  act: 0 = identity, 1 = ReLU.
 * tile (tuning / tests): 0 = automatic (7), 1 = generic 64x64 (any K), 2 = pipelined 64x64,
 * 3 = persistent pipelined 64x64, 4 = pipelined 128x64, 5 = pipelined 128x128, 6 = stream-K form of 3
 * (every block gets the same number of 64-wide K units; tiles cut between blocks are summed by the
 * owning block in a fixed order, so results are run-to-run deterministic), 7 = the stream-K kernel
 * with the launch shape (whole tiles or equal unit ranges) chosen per problem by a cost model: the
 * default on the path; 8 = unstaged split-K kernel for launches with few rows (one 32x32 tile per block,
 * K in {128, 256, 512, 1024} split over four or eight waves; 7 hands it every launch of at most 1024 rows);
 * 9 = pipelined 128x64 with 16-wide K slices (two blocks per CU; 7 hands it the plain projections of at
 * least 4096 rows whose tile count fills the resident block slots evenly), 10 = pipelined 128x128 with
 * 16-wide slices (measurement only), 11 = the LDS-DMA kernel (both operands by global_load_lds, transposed accumulators,
 * 64 x 128 tiles at three blocks per CU, whole tiles + remaining tiles cut into K pieces; K % 32 == 0, K >= 64, N % 4 == 0,
 * leading dimensions % 4, 16-byte aligned operands; 7 hands it the launches of at least FF_DMA_MIN_ROWS rows), 12 = the same
 * kernel with 64 x 64 tiles (round 6; up to four blocks per CU).
 * Kernels 6/7 keep an internal 8 MB workspace per (device, stream), allocated at the first launch on
 * that stream.
 * ------------------------------------------------------------------------------------------- */
int ff_gemm_f32(const float* A, int lda, const float* A2, int n_split,
                const float* W, int ldw, const float* bias,
                const float* residual, int ldr, float* C, int ldc,
                int M, int N, int K, int act, int tile, ff_stream_t stream);

/* Same product for `batch` independent problems in one launch: problem z uses A + z*stride_a
 * (and A2 + z*stride_a), W + z*stride_w, C + z*stride_c (strides in elements; bias is shared;
 * residual unsupported for batch > 1).  Used for the pointer logits, one weight matrix
 * (= the wireframe's edge embeddings) per wireframe. */
int ff_gemm_f32_batched(const float* A, int lda, const float* A2, int n_split,
                        const float* W, int ldw, const float* bias,
                        const float* residual, int ldr, float* C, int ldc,
                        int M, int N, int K, int act, int tile,
                        int batch, long long stride_a, long long stride_w, long long stride_c,
                        ff_stream_t stream);

/* The same product with the neighbouring LayerNorm folded in (what ff_decode uses; removes the standalone
 * LayerNorm launches between the projections of a decoder layer, reference transformer.py:242-253):
 *   ln_stats_out : the launch that PRODUCES a LayerNorm input x (out-proj / linear2 with their residual) also
 *                  leaves, per row and per 32-column segment, (mean, M2 = sum of squared deviations) of the
 *                  stored values: [M][N/32][2] floats.  N % 32 == 0.
 *   ln_stats_in  : the launch that CONSUMES LayerNorm(x) reads x itself as A and normalises every row while
 *                  staging it: a <- (a - mean) * rstd with (mean, rstd) merged from the ln_nseg = K/32 segment
 *                  statistics (Chan's update: no E[x^2] - mean^2 cancellation).  gamma / beta are NOT applied
 *                  here: the caller passes the folded weight W' = W diag(gamma) and bias' = bias + W beta
 *                  (ff_fold_layernorm_linear), so that act(LN(x) W^T + bias) = act(z W'^T + bias').
 *   row_table    : optional [*, ld_row_table] table added to columns < row_cols, row m taking line m / row_div:
 *                  (LN(x) + pos_j) W^T = LN(x) W^T + (pos W^T)_j for the position-major rows of the decoder
 *                  (j = m / sequences).  Needs ln_stats_in and no residual.
 * tile: 0 (automatic), 3, 6, 7, 8, 11 or 12; K % 64 == 0 and K >= 128 for the fused forms, K <= 512 with ln_stats_in. */
typedef struct ff_gemm_ln_desc {
  const float* A; int lda;
  const float* W; int ldw;
  const float* bias;
  const float* residual; int ldr;
  float* C; int ldc;
  int M, N, K, act, tile;
  const float* ln_stats_in; int ln_nseg; float ln_eps;
  const float* row_table; int ld_row_table, row_div, row_cols;
  float* ln_stats_out;
} ff_gemm_ln_desc;
int ff_gemm_f32_ln(const ff_gemm_ln_desc* desc, ff_stream_t stream);

/* Fold a LayerNorm's affine part (and a learned position table) into the Linear that follows it:
 *   Wf[n,k] = W[n,k] * gamma[k]            bf[n] = bias[n] + sum_k W[n,k] * beta[k]
 *   P[j,n]  = sum_k pos[j,k] * W[n,k]      for j < pos_rows, n < pos_cols   (skipped when pos == NULL)
 * W [N,K] (ld = ldw), Wf [N,K] contiguous, bf [N], P [pos_rows, pos_cols] contiguous.  Done once per weight
 * binding by the host side of ff_decode (faceformer_amd/hip/engine.py); products on the f32 MFMA GEMM. */
int ff_fold_layernorm_linear(const float* W, int ldw, int N, int K, const float* bias, const float* gamma,
                             const float* beta, const float* pos, int ldpos, int pos_rows, int pos_cols,
                             float* Wf, float* bf, float* P, ff_stream_t stream);

/* The same product evaluated on the bf16 matrix cores with fp32 accuracy ("3 x bf16"): every fp32
 * operand is split exactly into three bf16 terms and the six partial products of weight >= 2^-16 are
 * accumulated in fp32 (v_mfma_f32_32x32x16_bf16); what is dropped is below one fp32 rounding of the
 * product, so the error is that of an ordinary fp32 dot product (tests compare both with fp64).
 * W is given pre-split: ff_split_weight_bf16x3 writes its three planes once per weight tensor
 * (ff_split_weight_bytes(N, K) bytes; layout [3][K/16][N][16] bf16, i.e. a 16-wide K slice of 32 rows is
 * one contiguous 1 KB block); activations are split inside the kernel.  K % 32 == 0, K >= 64; n_split % 128 == 0;
 * N % 4 == 0, ldc / ldr % 4 == 0 and 16-byte aligned A / bias / residual / C.  Keeps an internal 24 MB partial-tile workspace per
 * (device, stream) (ff_gemm_prepare_stream allocates it ahead of time).  Same replaced call sites as ff_gemm_f32;
 * ff_decode uses it for the decoder projections of launches with at least ff_decode_params.x3_min_rows rows. */
size_t ff_split_weight_bytes(int N, int K);
int ff_split_weight_bf16x3(const float* W, int ldw, int N, int K, void* planes, ff_stream_t stream);
int ff_gemm_x3(const float* A, int lda, const float* A2, int n_split, const void* w_planes,
               const float* bias, const float* residual, int ldr, float* C, int ldc,
               int M, int N, int K, int act, ff_stream_t stream);

/* The 3 x bf16 product with the neighbouring LayerNorm folded in: the descriptor of ff_gemm_f32_ln (same meaning of
 * ln_stats_in / row_table / ln_stats_out; reference transformer.py:242-253), the weight given as the planes of the
 * FOLDED weight (ff_fold_layernorm_linear, then ff_split_weight_bf16x3); desc->W / ldw / tile are ignored.  The planes
 * describe a [plane_rows, K] weight of which the product uses rows [row0, row0 + N) (plane_rows = 0: exactly N rows).
 * w_colsum (optional, with ln_stats_in): [plane_rows] row sums of the folded weight -- the normalisation is then applied in the
 * epilogue, LN(x) W'^T = rstd (x W'^T - mean colsum), and the K loop runs at the plain kernel's speed; its rounding error
 * carries the factor (1 + |mean| / sigma) of the row (ff_gemm_x3.hip: x3_ln_linear), NULL = rows normalised before the product.
 * ln_stats_in needs K = 512 (ln_nseg = 16).  N % 4 == 0, ldc / ldr / ld_row_table % 4 == 0, row_cols % 4 == 0,
 * 16-byte aligned operands (the kernel moves 16-byte pieces).  ff_decode uses it on the steps that take the 3 x bf16
 * projections, so that those steps launch no stand-alone LayerNorm either. */
int ff_gemm_x3_ln(const ff_gemm_ln_desc* desc, const void* w_planes, int plane_rows, int row0, const float* w_colsum,
                  ff_stream_t stream);

/* "2 x fp16" (round 6): the same products on the fp16 matrix cores with HALF the partial products.  x = x1 + x2, x1 = fp16(x),
 * x2' = fp16((x - x1) 2^11) (the second term is kept at 2^11 times its value: the magnitude of the first, outside fp16's
 * subnormals): 22 mantissa bits per operand, THREE products x1 y1 + (x1 y2' + x2' y1) 2^-11 (v_mfma_f32_32x32x16_f16; x1 y1 in
 * its own accumulator).  What is dropped (x2 y2, the last two bits of each operand) stays below the rounding error an fp32
 * dot product of that length accumulates anyway: against fp64 the result is as close as ff_gemm_f32's (op tests compare the three;
 * emulation: profiles/r06/fp16_split_error_table.txt).  fp16 has five exponent bits: every |a| must be < 65504 (weights: checked
 * by the caller when it makes the planes).  LayerNorm-normalised rows are bounded by sqrt(K); the epilogue form (w_colsum) feeds
 * the RAW rows at 2^-6 (|x| < 4.2e6, exact scaling).  Planes: ff_split_weight_fp16x2, [2][K/16][N][16] fp16,
 * ff_split_weight_fp16x2_bytes(N, K) bytes.  Arguments, restrictions and replaced call sites as ff_gemm_x3 / ff_gemm_x3_ln. */
size_t ff_split_weight_fp16x2_bytes(int N, int K);
int ff_split_weight_fp16x2(const float* W, int ldw, int N, int K, void* planes, ff_stream_t stream);
int ff_gemm_x2h(const float* A, int lda, const float* A2, int n_split, const void* w_planes,
                const float* bias, const float* residual, int ldr, float* C, int ldc,
                int M, int N, int K, int act, ff_stream_t stream);
int ff_gemm_x2h_ln(const ff_gemm_ln_desc* desc, const void* w_planes, int plane_rows, int row0, const float* w_colsum,
                   ff_stream_t stream);

/* Launch shape of the 3 x bf16 kernel (process-wide; tests and tools/): 0 = the default -- whole tiles, and when the tile
 * count is not a multiple of the CU count the remaining tiles cut into 2 / 4 / 8 K-pieces, one piece per CU, summed by the
 * block that holds the tile's last piece in ascending piece order --, 1 = whole tiles only, 2 = equal K-unit ranges per
 * block (every tile that straddles two blocks exchanged the same way). */
int ff_set_x3_tuning(int shape);

/* Launch-shape tuning of the stream-K kernel (process-wide; tests and tools/): a block is never handed
 * fewer than `min_units` K units (of 64); launches with at least `two_per_cu_units` units use 512
 * blocks (two per CU), smaller ones at most 256; tile 7 cuts tiles only when that saves more than
 * `fix_tenths`/10 units of per-CU work; launches of at most `small_max_rows` rows (all problems of a
 * batched call together) go to the unstaged split-K kernel (0: never). */
int ff_set_gemm_tuning(int min_units, int two_per_cu_units, int fix_tenths, int small_max_rows);

/* Allocate the partial-tile exchange buffers of the stream-K kernels (f32 and 3 x bf16) for (current device, stream) now
 * instead of at the first launch (hipMalloc + device synchronisation: keep it out of timed or captured
 * regions).  ff_decode calls it for the caller's stream and its internal streams before enqueuing. */
int ff_gemm_prepare_stream(ff_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * G4/G5/G6  Multi-head attention core: softmax(q k^T * scale + mask) v for `num_groups` groups
 * that each share one key/value set, `num_heads` heads of FF_HEAD_DIM columns.  Replaces the
 * q*scale / baddbmm / softmax / bmm chain of torch's multi_head_attention_forward at the call
 * sites transformer.py:170-171 (encoder self), 244-245 (decoder self), 248-251 (decoder cross).
 * Flash-style: key tiles staged in LDS, scores on the f32 MFMA, online softmax in registers.
 *
 * Row addressing (lets one kernel serve position-major decoder tensors without copies):
 *   query/output row of (group g, query i) = g*q_group_stride + (i / q_inner)*q_outer_stride
 *                                            + (i % q_inner)
 *   key/value   row of (group g, key j)    = g*k_group_stride + j*k_stride
 * Head h reads columns [h*64, h*64+64) of q/k/v and writes the same columns of o.
 * Masking: key j of group g is ignored when j >= kv_len[g] (kv_len NULL: nk), when
 * key_mask[g*mask_stride + j] != 0 (key_mask NULL: none) or, if causal != 0, when j > i.
 * A query whose keys are all masked yields NaN in torch; here it yields 0 (never happens on the
 * path: the four special-token keys are never masked, reference model.py:61-64).
 * ------------------------------------------------------------------------------------------- */
typedef struct ff_attn_desc {
  const float* q; const float* k; const float* v; float* o;
  int ldq, ldk, ldv, ldo;
  int num_groups, num_heads;
  int nq;                 /* queries per group */
  int q_group_stride, q_inner, q_outer_stride;
  int nk;                 /* max keys per group */
  int k_group_stride, k_stride;
  const int* kv_len;      /* [num_groups] or NULL */
  const unsigned char* key_mask; int mask_stride; /* [num_groups, mask_stride] or NULL; 1 = masked */
  int causal;
  float scale;            /* 1/sqrt(64) = 0.125 on the path */
  const void* kv_planes;  /* optional (round 6): the fp16 planes of k / v made by ff_attention_split_kv for exactly these groups, heads
                             and nk -- launches that qualify (nk <= 288, no causal mask, enough query tiles) then run on the fp16 matrix
                             cores with fp32 accuracy ("2 x fp16": ff_attention_x2h.hip); NULL: the f32 kernels */
} ff_attn_desc;

/* K | V of every (group, head) pair -> two fp16 planes each, in the layout the 2 x fp16 attention kernel copies into LDS
 * (ff_attention_planes_bytes(num_groups, num_heads) bytes: 148 480 per pair).  k / v / ldk / ldv / nk / k_group_stride / k_stride as in
 * ff_attn_desc; 1 <= nk <= 288.  Made once per batch and layer by ff_decode (cross-attention keys are the encoder memory:
 * reference transformer.py:248-250); every |k|, |v| must be below 65504 (fp16's range). */
size_t ff_attention_planes_bytes(int num_groups, int num_heads);
int ff_attention_split_kv(const float* k, const float* v, int ldk, int ldv, int num_groups, int num_heads, int nk,
                          int k_group_stride, int k_stride, void* planes, ff_stream_t stream);

int ff_attention(const ff_attn_desc* desc, ff_stream_t stream);
/* Kernel selection for ff_attention (tuning / tests): 0 automatic, 1 block-shared LDS staging,
 * 2 wave-independent with in-block key splitting, 3 K/V-resident (key sets of at most 288 rows without a causal mask;
 * other launches fall back to the automatic choice between 1 and 2), 4 the 2 x fp16 kernel whenever the descriptor carries planes
 * and the launch is inside its limits (automatic: from the K/V-resident kernel's threshold on).  Returns the previous value.
 * The K/V-resident kernel parks partial (max, sum, O) records in the stream's scratch area (the one ff_gemm_prepare_stream
 * allocates: 24 MB per (device, stream), shared with the projection kernels of that stream in stream order). */
int ff_set_attention_algo(int algo);

/* ---------------------------------------------------------------------------------------------
 * General attention core (round 6): the part of nn.MultiheadAttention's surface the 64-wide MFMA kernels above do not take --
 * ANY head width (reference transformer.py:131,191-192 pass num_model / num_head through; every reference config gives 64) and
 * torch's `attn_mask` in its general forms: the `mask` / `src_mask` of the encoder (transformer.py:70-73,164-176), a `tgt_mask`
 * that is not the causal triangle and the `memory_mask` of the decoder (transformer.py:95-101,235-256).  Same row addressing,
 * kv_len / key_mask / causal meaning as ff_attn_desc; head h reads columns [h*head_dim, (h+1)*head_dim).
 *   attn_bias : additive fp32 mask, element (query i, key j) at attn_bias[b*attn_batch_stride + i*attn_ld + j] with
 *               b = group*num_heads + head (torch's [N*H, L, S] order); attn_batch_stride 0 = one [nq, nk] matrix for all
 *   attn_mask : boolean mask (1 = the key is removed), same addressing (in bytes); either, both or neither may be given
 * Arithmetic of torch's explicit-weights path (what the reference calls): q*scale first, scores, + bias, -inf for removed keys,
 * softmax, P V.  A query with no key left yields NaN like torch (ff_attention yields 0).  One wavefront per (group, head, query)
 * on the VALU: a correctness surface for module users, not the decode path's kernel.  head_dim + nk <= ~10 000 floats (LDS).
 * ------------------------------------------------------------------------------------------- */
typedef struct ff_attn_general_desc {
  const float* q; const float* k; const float* v; float* o;
  int ldq, ldk, ldv, ldo;
  int num_groups, num_heads, head_dim;
  int nq;
  int q_group_stride, q_inner, q_outer_stride;
  int nk;
  int k_group_stride, k_stride;
  const int* kv_len;
  const unsigned char* key_mask; int mask_stride;
  int causal;
  const float* attn_bias;
  const unsigned char* attn_mask;
  int attn_ld;
  long long attn_batch_stride;
  float scale;            /* head_dim ** -0.5 for nn.MultiheadAttention */
} ff_attn_general_desc;
int ff_attention_general(const ff_attn_general_desc* desc, ff_stream_t stream);

/* Tuning knobs (round 6): every A/B switch of the library is one int in one table.  `name` is the environment
 * variable that initialises the knob when the library is first used (FF_L0_FOLD, FF_POINTER_FOLD, FF_LAST_QKV_ONE_LAUNCH_ROWS,
 * FF_PINNED_COUNTERS, FF_DEBUG_TIMING, FF_DMA_MIN_ROWS, FF_DMA_MIN_ROWS_N512, FF_DMA_MIN_ROWS_WIDE, FF_SK_HYBRID, FF_SK_HYBRID_FIX,
 * FF_SK_HYBRID_MAXLEFT8, FF_SK_HYBRID_MINU, FF_SK_HYBRID_FORCE, FF_NO_PANEL, FF_X3_SMALL_SPLIT, FF_RK_SPLIT_OLD, FF_RK_SPLIT_YOUNG,
 * FF_RK_PHASE, FF_RK_ROTATE).  ff_set_tuning changes it for the process (tests and tools flip knobs without child processes);
 * a decode takes ONE snapshot of the knobs that shape it when it starts.  ff_reset_tuning restores the built-in defaults.
 * Returns FF_ERR_ARG for an unknown name.  The defaults are the product; no reference interface corresponds to these. */
int ff_set_tuning(const char* name, int value);
int ff_get_tuning(const char* name, int* value);
int ff_reset_tuning(void);

/* ---------------------------------------------------------------------------------------------
 * G7/G8/G9  Pointer head: logits of every sequence against the edge embeddings of its wireframe,
 * padding mask, argmax, and the feedback gather of the chosen embedding row.  Replaces
 * select_next (reference model_para.py:173-179, model.py:161-167), the per-step torch.gather
 * (model_para.py:217-219) and the stop-rule reductions (model_para.py:232, model.py:207).
 *   logit[b,s] = < memory[w(b), s, :], p[b, :] >,  w(b) = b / seqs_per_group
 *   masked (mask[w,s] != 0, s >= kv_len[w], or extra_mask[b,s] != 0) -> -FLT_MAX (finfo.min, not -inf)
 *   next_tok[b] = argmax_s (lowest index on ties)
 * Two code paths, same results up to fp32 summation order:
 *   logits == NULL : streaming kernel, one wavefront per sequence; p in registers, embedding rows
 *                    streamed with coalesced float4 loads, 64-lane butterfly reduction per logit;
 *   logits != NULL : [B, ldlogits] scratch/output: raw logits by the batched f32-MFMA GEMM (one
 *                    problem per wireframe: [seqs_per_group, E] x [E, S]), then one wavefront per
 *                    sequence masks its row in place and reduces (value, index) pairs by shuffles
 *                    (what the engine uses: 40x faster at 256 sequences per wireframe).
 * Optional outputs (NULL to skip):
 *   best/second [B]   top-2 logits (parity margins),  logits [B, ldlogits] masked logits,
 *   next_rows [B, ldnext] = memory[w(b), next_tok[b], :]  (next decoder input row),
 *   count_lt / count_eq: *count_lt += #{b: next_tok[b] >= lt_bound}, *count_eq += #{b: next_tok[b] == eq_value}
 *   (device counters for the stop rules; caller zeroes them).
 * ------------------------------------------------------------------------------------------- */
int ff_pointer_argmax(const float* p, int ldp, const float* memory, int S, int E,
                      const unsigned char* mask, const int* kv_len,
                      const unsigned char* extra_mask, int ldextra,
                      int B, int seqs_per_group,
                      int* next_tok, float* best, float* second, float* logits, int ldlogits,
                      float* next_rows, int ldnext,
                      int* count_ge, int ge_bound, int* count_eq, int eq_value,
                      ff_stream_t stream);

/* out[b,:] = memory[(b / seqs_per_group), tok[b], :]   (first decoder input: anchors / SOS;
 * reference model_para.py:217-219 at step 0). */
int ff_gather_rows(const float* memory, int S, int E, const int* tok, int B, int seqs_per_group,
                   float* out, int ldout, ff_stream_t stream);

/* value_embed[n, 0:num_token, :] = tok_embed ; value_embed[n, num_token + l, :] = edge_embed[n*L + l, :]
 * (the torch.cat of reference embedding.py:36). */
int ff_assemble_embedding(const float* tok_embed, int num_token, const float* edge_embed, int ld_edge,
                          int N, int L, int E, float* out, ff_stream_t stream);

/* process_masks + key lengths in one launch: mask_out[n, 0:num_token] = 0 (special tokens are never masked, reference
 * model.py:61-69 / model_para.py:62-70), mask_out[n, num_token + l] = (input_mask[n, l] != 0); kv_len[n] = 1 + the last
 * unmasked key of wireframe n (0: every key masked).  input_mask: N x L bytes (a torch.bool tensor's storage), 1 = padding. */
int ff_prepare_mask(const unsigned char* input_mask, int N, int L, int num_token, unsigned char* mask_out, int* kv_len,
                    ff_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-path engine.  Weights are the tensors of the reference state_dict,
 * fp32, on the device, passed by pointer -- no repacking.
 * ------------------------------------------------------------------------------------------- */
typedef struct ff_mha_weights {
  const float* in_proj_w;  /* [3E, E]  rows: Wq | Wk | Wv */
  const float* in_proj_b;  /* [3E] */
  const float* out_w;      /* [E, E] */
  const float* out_b;      /* [E] */
} ff_mha_weights;

typedef struct ff_layer_weights {
  ff_mha_weights self_attn;
  ff_mha_weights cross_attn;           /* decoder layers only */
  const float *lin1_w, *lin1_b;        /* [FF, E], [FF] */
  const float *lin2_w, *lin2_b;        /* [E, FF], [E] */
  const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *norm3_w, *norm3_b; /* [E]; norm3: decoder */
  /* optional (decoder layers): ff_split_weight_bf16x3 planes of self_attn.in_proj_w, lin1_w, lin2_w; when
     given, ff_decode evaluates these projections with ff_gemm_x3 on steps that have at least
     ff_decode_params.x3_min_rows prefix rows */
  const void *in_proj_planes, *lin1_planes, *lin2_planes;
  /* the same for self_attn.out_w, the q rows of cross_attn.in_proj_w ([E, E]) and cross_attn.out_w */
  const void *self_out_planes, *cross_q_planes, *cross_out_planes;
  /* optional (decoder layers), written by ff_fold_layernorm_linear: the affine part of norm1 / norm2 / norm3 and
     the query-position table folded into the projection that consumes the LayerNorm (FF_FUSE_LAYERNORM):
       ln1_* : norm1 -> self_attn.in_proj   Wf [3E,E], bf [3E], P = qpos [Wq;Wk]^T [qpos_len, 2E]
       ln2_* : norm2 -> q rows of cross_attn.in_proj   Wf [E,E], bf [E], P = qpos Wq^T [qpos_len, E]
       ln3_* : norm3 -> linear1             Wf [FF,E], bf [FF] */
  const float *ln1_w, *ln1_b, *ln1_pos;
  const float *ln2_w, *ln2_b, *ln2_pos;
  const float *ln3_w, *ln3_b;
  /* optional: ff_split_weight_bf16x3 planes of the FOLDED weights ln1_w [3E,E], ln2_w [E,E], ln3_w [FF,E]: with them (and the
     planes of the out-proj / linear2 weights above) the steps that take the 3 x bf16 projections keep the LayerNorm folding
     (ff_gemm_x3_ln) at every size */
  const void *ln1_planes, *ln2_planes, *ln3_planes;
  /* optional: row sums of ln1_w / ln2_w / ln3_w ([3E], [E], [FF]): with them ff_decode's split-product steps apply the
     LayerNorm in the consumer's epilogue (ff_gemm_x3_ln, w_colsum) */
  const float *ln1_csum, *ln2_csum, *ln3_csum;
} ff_layer_weights;

typedef struct ff_model {
  int E, H, FF, num_enc_layers, num_dec_layers;
  int in_dim;              /* num_points_per_line * point_dim (100) */
  int num_token;           /* 4 special tokens */
  int pos_len, qpos_len;   /* rows of the two learned position tables */
  float ln_eps;
  const float* tok_embed;              /* val_enc.embedding_token.weight [num_token, E] */
  const float *emb_w1, *emb_b1;        /* val_enc.embedding_value.0  [E, in_dim], [E] */
  const float *emb_w2, *emb_b2;        /* val_enc.embedding_value.2  [E, E], [E] */
  const float* pos_table;              /* pos_enc.pos_embed.weight [pos_len, E] */
  const float* qpos_table;             /* query_pos_enc.pos_embed.weight [qpos_len, E] */
  ff_layer_weights enc[FF_MAX_LAYERS];
  const float *enc_norm_w, *enc_norm_b;
  ff_layer_weights dec[FF_MAX_LAYERS];
  const float *dec_norm_w, *dec_norm_b;
  const float *proj_w, *proj_b;        /* project [E, E], [E] */
  const float *proj_fold_w, *proj_fold_b; /* optional: decoder.norm folded into project (ff_fold_layernorm_linear) */
  int split_kind;          /* what the `*_planes` of the decoder layers hold: 0 = three bf16 planes (ff_split_weight_bf16x3, six
                              products per fp32 product), 1 = two fp16 planes (ff_split_weight_fp16x2, three products; round 6) */
} ff_model;

/* Encoder (a1-a4 of SURVEY.md 8a): embedding MLP + token rows, 6 pre-norm layers, final LayerNorm.
 *   input  [N, L, in_dim]     edge polylines (flattened points)
 *   mask   [N, S] uint8       S = L + num_token, 1 = padding key (after process_masks)
 *   kv_len [N]    int32       1 + index of the last unmasked key of each wireframe
 *   memory [N, S, E]          output
 * Replaces val_enc + encoder (reference model_para.py:194,210; transformer.py:70-83). */
size_t ff_encode_workspace_bytes(const ff_model* m, int N, int L);
int ff_encode(const ff_model* m, const float* input, const unsigned char* mask, const int* kv_len,
              int N, int L, float* memory, void* workspace, size_t workspace_bytes,
              ff_stream_t stream);

enum ff_variant { FF_PARALLEL = 0, FF_SEQ2SEQ = 1 };
enum ff_decode_flags {
  FF_REUSE_LAYER0_QKV = 1,   /* layer-0 self-attention q,k,v computed once per filled position */
  FF_LAST_LAYER_LAST_ROW = 2,/* last decoder layer evaluated for the newest position only */
  FF_RETURN_POINTER = 4,     /* also produce project(decoder(...)) for ALL prefix rows of the last step */
  FF_NO_STOP = 8,            /* run all T-1 steps and do not apply the stop rule (multi-GPU: the caller
                                all-reduces step_counts and applies the GLOBAL rule) */
  FF_FUSE_LAYERNORM = 32,    /* decoder: no standalone LayerNorm launches between the projections -- the GEMM that
                                produces a LayerNorm input leaves per-row segment statistics, the GEMM that consumes
                                it normalises its A rows while staging them (ff_gemm_f32_ln).  Needs the folded
                                weights (ln1_w ... proj_fold_b) and E, FF multiples of 64, 128 <= E <= 512; otherwise ignored */
  FF_DEDUP_PAD_ANCHORS = 16, /* parallel variant: the F - num_input[w] padding-anchor sequences of a wireframe
                                (start token num_token-1, reference model_para.py:204-205) are identical by
                                construction; decode ONE of them and copy its tokens into all those rows of
                                `predict`.  Needs num_input_host; ignored when an extra mask is given */
  FF_NO_L0_FOLD = 1024,      /* this call: the LayerNorm of the rows a step appends as its own launch (the pointer launch leaves no
                                statistics); as tuning knob FF_L0_FOLD = 0, but per call -- it changes the workspace layout, so
                                ff_decode_workspace_bytes must see the same flags */
  FF_NO_POINTER_FOLD = 2048, /* this call: decoder.norm + project and the pointer's dot products as two launches also for
                                one-wireframe micro-batches (knob FF_POINTER_FOLD = 0, per call; changes the workspace layout) */
  FF_STOP_EACH_EOS = 512     /* seq2seq variant: the per-step counter counts a sequence's FIRST EOS only, so the cumulative
                                rule "count == N" fires at the first step by which EVERY wireframe has produced an EOS -- the
                                rule a caller needs when the records of a batch must equal those of one-wireframe decodes
                                (the reference's rule, model.py:207-210, counts repeated EOS of one sample as well and can stop
                                a batch before another sample has produced its own).  Not a reference behaviour: off by default */
};
/* (64 / 128 / 256 were FF_CHAIN / FF_FLOW / FF_GRAPH, the persistent-launch experiments of round 3: built, parity-tested on
   every golden, measured slower than launch-per-operator three ways -- DESIGN.md 8 -- and removed in round 5.) */

/* External stop rule (multi-GPU: SURVEY.md 8e).  Called on the HOST, from inside ff_decode, every sync_every steps with this
 * call's per-step counters of steps [0, num_steps) -- #{tokens >= num_token} (parallel) or #{tokens == EOS} (seq2seq) over the
 * sequences THIS call decodes; num_steps runs one period behind the steps already enqueued (the GPU never drains).  A
 * non-zero return ends the decode after the steps enqueued so far.  A sharded caller sums the counters over its ranks in
 * here (a small host-side all-reduce) and applies the reference's rule to the batch-global numbers. */
typedef int (*ff_stop_fn)(void* user, const int* step_counts, int num_steps);

typedef struct ff_decode_params {
  int variant;          /* ff_variant */
  int N;                /* wireframes */
  int L;                /* padded edges per wireframe; S = L + num_token */
  int F;                /* sequences per wireframe: max(num_input) (parallel) or 1 (seq2seq) */
  int T;                /* max_face_length / label_seq_length; at most T-1 decode steps */
  int chunk_wireframes; /* wireframes per micro-batch (<=0: all) */
  int chunk_seqs;       /* >0 and < F: additionally split every wireframe into groups of this many
                           sequences (sequences are independent; groups decode concurrently) */
  int num_streams;      /* micro-batches are issued round-robin on this many HIP streams (internal
                           pool, forked from / joined into `stream`); <=1: everything on `stream` */
  int sync_every;       /* evaluate the stop rule on the host every k steps (<=0: only at the end) */
  int flags;            /* ff_decode_flags */
  int tok_sos, tok_eos; /* seq2seq start / stop tokens */
  int x3_min_rows;      /* > 0: decoder projections whose weight planes are bound (q|k|v, linear1, linear2)
                           run on the bf16 matrix cores (3 x bf16 split, fp32 accuracy) when the micro-batch
                           has at least this many prefix rows (t * sequences); 0: never */
  int chunk_max_seqs;   /* > 0: a micro-batch of several wireframes holds at most this many sequences
                           (a single wireframe is never cut by it); 0: no limit.  seq2seq (one sequence per wireframe):
                           > 0 REPLACES chunk_wireframes as the micro-batch size */
  int ln_fuse_max_rows; /* FF_FUSE_LAYERNORM applies to decode steps with at most this many active rows
                           (t * sequences of the micro-batch); 0: the default (12288; x3_min_rows - 1 when the 3 x bf16
                           projections are in use: the steps that take them launch their LayerNorms) */
  ff_stop_fn stop_fn;   /* optional: replaces the LOCAL stop rule (needs sync_every > 0; ignored with FF_NO_STOP).
                           `predict` then keeps every step that was executed (as with FF_NO_STOP): the caller zero-pads
                           after the step its global rule names */
  void* stop_user;      /* first argument of stop_fn */
} ff_decode_params;

/* Greedy pointer decode (a5-a12 of SURVEY.md 8a).
 *   memory, mask, kv_len : as produced by / given to ff_encode
 *   num_input [N] int32   : DEVICE, real edge count per wireframe (parallel anchors, model_para.py:201-207)
 *   num_input_host [N]    : the same values on the HOST (plans the micro-batches and the padding-anchor
 *                           de-duplication); NULL: every wireframe decodes all F sequences
 *   extra_mask            : optional [B, S] uint8 additional pointer mask (co-edge style), or NULL
 *   predict [N*F, T] int64: output tokens incl. the start token, zero padded after the stop step
 *   steps_done            : host int, number of decode steps the reference semantics executed
 *   step_counts           : optional host int[T-1]: per executed step, #{tokens >= num_token} (parallel)
 *                           or #{tokens == EOS} (seq2seq) over the DECODED sequences -- the inputs of the
 *                           stop rules (with de-duplication a padding-anchor sequence counts once: the
 *                           parallel rule only tests the count against zero)
 *   pointer_out           : optional [steps_done, N*F, E] (FF_RETURN_POINTER), position-major
 *   trace_logits          : optional [T-1, N*F, S] masked logits of every step (tests), or NULL; within a step
 *                           the entries are indexed by the COMPACT sequence id (see seq_of_row; only the
 *                           first Bd <= N*F entries of a step are written)
 *   trace_best/second     : optional [T-1, N*F] top-2 logits, same indexing, or NULL
 *   seq_of_row            : optional DEVICE int[N*F]: compact sequence id behind every row of `predict`
 * Stop rules reproduced exactly: parallel = first step whose tokens are all < num_token
 * (model_para.py:232); seq2seq = cumulative EOS count == N (model.py:207-210). */
size_t ff_decode_workspace_bytes(const ff_model* m, const ff_decode_params* p, const int* num_input_host);
int ff_decode(const ff_model* m, const ff_decode_params* p,
              const float* memory, const unsigned char* mask, const int* kv_len,
              const int* num_input, const int* num_input_host, const unsigned char* extra_mask,
              int64_t* predict, int* steps_done, int* step_counts, float* pointer_out,
              float* trace_logits, float* trace_best, float* trace_second, int* seq_of_row,
              void* workspace, size_t workspace_bytes, ff_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FACEFORMER_HIP_H */
