// General multi-head attention core (gfx950): any head width, additive / boolean attention masks.
//
// The decode path of every reference config runs 64-wide heads with key-padding masks only (ff_attention.hip's three MFMA
// kernels).  The reference's blocks accept more than that (faceformer/transformer.py:131,191-192: any num_model / num_head;
// :70-73,95-101: `mask` / `src_mask`, `tgt_mask`, `memory_mask` of torch's nn.MultiheadAttention -- boolean or additive float,
// [nq, nk] or one matrix per (batch, head)); this kernel is that remainder of the module surface.  It is a latency-insensitive
// VALU kernel by design: one wavefront per (group, head, query), scores with one key per lane, softmax through wave
// reductions, P V with one output column per lane (key-interleaved lane groups for heads narrower than 64 columns).
//
// Arithmetic follows torch's explicit (need_weights=True) path, which is what the reference calls: q is scaled first,
// scores = q_scaled k^T (+ additive mask, -inf where a boolean mask / key padding / kv_len / causal rule removes a key),
// softmax, P V.  A query with no key left yields NaN, as torch's softmax over an all -inf row does (the 64-wide fast kernels
// yield 0 there; the case never occurs on the decode path).
#include <math.h>

#include <atomic>

#include "ff_common.h"

namespace {

constexpr int GA_WAVES = 4;   // wavefronts per block, one (group, head, query) unit each

struct GAParams {
  ff_attn_general_desc d;
  long long units;   // num_groups * num_heads * nq
  int d_pad;         // head_dim rounded up to 4 (LDS slot of q)
  int row_floats;    // LDS floats per wave: d_pad + nk rounded up to 4 (16-byte aligned slots)
  int lane_groups;   // key-interleaved lane groups of the P V pass (head_dim 4 / 8 / 16 / 32: 16 / 8 / 4 / 2, else 1)
};

__global__ __launch_bounds__(GA_WAVES* FF_WAVE) void attention_general_kernel(GAParams p) {
  extern __shared__ float lds[];
  const ff_attn_general_desc& d = p.d;
  const int wave = threadIdx.x / FF_WAVE, lane = threadIdx.x % FF_WAVE;
  float* qs = lds + (size_t)wave * p.row_floats;   // [d_pad] scaled query, then [nk] scores / probabilities
  float* sc = qs + p.d_pad;
  const int hd = d.head_dim;
  const float ninf = -INFINITY;
  for (long long unit = (long long)blockIdx.x * GA_WAVES + wave; unit < p.units; unit += (long long)gridDim.x * GA_WAVES) {
    const int i = (int)(unit % d.nq);
    const long long gh = unit / d.nq;
    const int h = (int)(gh % d.num_heads), g = (int)(gh / d.num_heads);
    const long long qrow = (long long)g * d.q_group_stride + (long long)(i / d.q_inner) * d.q_outer_stride + (i % d.q_inner);
    const float* qp = d.q + qrow * d.ldq + (long long)h * hd;
    for (int c = lane; c < hd; c += FF_WAVE) qs[c] = qp[c] * d.scale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the query is in LDS before any lane reads it back (one wave: LDS is in order)
    int nk = d.nk;
    if (d.kv_len) nk = min(nk, max(d.kv_len[g], 0));
    const unsigned char* km = d.key_mask ? d.key_mask + (long long)g * d.mask_stride : nullptr;
    const long long mrow = (d.attn_batch_stride ? gh * d.attn_batch_stride : 0) + (long long)i * d.attn_ld;
    const float* ab = d.attn_bias ? d.attn_bias + mrow : nullptr;
    const unsigned char* am = d.attn_mask ? d.attn_mask + mrow : nullptr;
    // scores: one key per lane, the key's head slice read as 16-byte pieces when it is aligned that way
    const bool vec = (hd % 4 == 0) && (d.ldk % 4 == 0) && ((reinterpret_cast<uintptr_t>(d.k) & 15u) == 0);
    float mx = ninf;
    for (int j = lane; j < d.nk; j += FF_WAVE) {
      float s = ninf;
      const bool live = j < nk && !(km && km[j]) && !(d.causal && j > i) && !(am && am[j]);
      if (live) {
        const float* kp = d.k + ((long long)g * d.k_group_stride + (long long)j * d.k_stride) * d.ldk + (long long)h * hd;
        float acc = 0.f;
        if (vec) {
          for (int c = 0; c < hd; c += 4) {
            const f32x4 kv = *reinterpret_cast<const f32x4*>(kp + c);
            const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + c);
            acc = fmaf(qv[0], kv[0], acc);
            acc = fmaf(qv[1], kv[1], acc);
            acc = fmaf(qv[2], kv[2], acc);
            acc = fmaf(qv[3], kv[3], acc);
          }
        } else {
          for (int c = 0; c < hd; ++c) acc = fmaf(qs[c], kp[c], acc);
        }
        s = ab ? acc + ab[j] : acc;
      }
      sc[j] = s;
      mx = fmaxf(mx, s);
    }
    mx = ff_wave_max(mx);
    float sum = 0.f;
    if (mx > ninf) {   // (an additive mask may hold -inf as well: those keys end at exp(-inf) = 0 like the boolean ones)
      for (int j = lane; j < d.nk; j += FF_WAVE) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
      }
      sum = ff_wave_sum(sum);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* op = d.o + qrow * d.ldo + (long long)h * hd;
    if (!(mx > ninf) || !(sum > 0.f) || sum != sum) {
      // no key left (or a NaN / +inf score): torch's softmax row is NaN, and so is its product with V
      for (int c = lane; c < hd; c += FF_WAVE) op[c] = NAN;
    } else {
      const float inv = 1.f / sum;
      const int G = p.lane_groups;
      if (G > 1) {   // hd in {4, 8, 16, 32}: lane = grp * hd + c; group grp takes keys grp, grp + G, ...
        const int c = lane % hd, grp = lane / hd;
        float acc = 0.f;
        for (int j = grp; j < d.nk; j += G) {
          const float pj = sc[j];
          if (pj != 0.f)
            acc = fmaf(pj, d.v[((long long)g * d.k_group_stride + (long long)j * d.k_stride) * d.ldv + (long long)h * hd + c], acc);
        }
        for (int off = hd; off < FF_WAVE; off <<= 1) acc += __shfl_xor(acc, off, FF_WAVE);
        if (grp == 0) op[c] = acc * inv;
      } else {
        for (int c0 = 0; c0 < hd; c0 += FF_WAVE) {
          const int c = c0 + lane;
          float acc = 0.f;
          if (c < hd) {
            for (int j = 0; j < d.nk; ++j) {
              const float pj = sc[j];
              if (pj != 0.f)
                acc = fmaf(pj, d.v[((long long)g * d.k_group_stride + (long long)j * d.k_stride) * d.ldv + (long long)h * hd + c], acc);
            }
            op[c] = acc * inv;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the LDS slot is reused by this wave's next unit
  }
}

}  // namespace

extern "C" int ff_attention_general(const ff_attn_general_desc* desc, ff_stream_t stream) {
  FF_CHECK_ARG(desc != nullptr, "ff_attention_general: null descriptor");
  GAParams p;
  p.d = *desc;
  const ff_attn_general_desc& d = p.d;
  if (d.num_groups == 0 || d.nq == 0) return FF_OK;
  FF_CHECK_ARG(d.num_groups > 0 && d.num_heads > 0 && d.nq > 0 && d.nk >= 0, "ff_attention_general: bad counts");
  FF_CHECK_ARG(d.head_dim >= 1 && d.head_dim <= 1024, "ff_attention_general: head_dim %d outside 1..1024", d.head_dim);
  FF_CHECK_ARG(d.q && d.o && (d.nk == 0 || (d.k && d.v)), "ff_attention_general: null tensor");
  const long long width = (long long)d.num_heads * d.head_dim;
  FF_CHECK_ARG(d.ldq >= width && d.ldo >= width && (d.nk == 0 || (d.ldk >= width && d.ldv >= width)),
               "ff_attention_general: ld smaller than num_heads*head_dim");
  FF_CHECK_ARG(d.q_inner > 0, "ff_attention_general: q_inner must be positive");
  FF_CHECK_ARG(!(d.attn_bias || d.attn_mask) || d.attn_ld >= d.nk, "ff_attention_general: attn_ld smaller than nk");
  FF_CHECK_ARG(d.attn_batch_stride >= 0, "ff_attention_general: negative attn_batch_stride");
  FF_CHECK_ARG(!d.key_mask || d.mask_stride >= d.nk, "ff_attention_general: mask_stride smaller than nk");
  p.units = (long long)d.num_groups * d.num_heads * d.nq;
  p.d_pad = (d.head_dim + 3) / 4 * 4;
  p.lane_groups = (d.head_dim == 4 || d.head_dim == 8 || d.head_dim == 16 || d.head_dim == 32) ? FF_WAVE / d.head_dim : 1;
  p.row_floats = p.d_pad + (d.nk + 3) / 4 * 4;
  const size_t lds_bytes = (size_t)GA_WAVES * p.row_floats * sizeof(float);
  FF_CHECK_ARG(lds_bytes <= 160 * 1024, "ff_attention_general: head_dim + nk = %d + %d exceeds the 160 KB of LDS of a CU (4 rows of %d floats)",
               d.head_dim, d.nk, p.row_floats);
  hipStream_t st = (hipStream_t)stream;
  if (lds_bytes > 64 * 1024) {
    static std::atomic<size_t> attr_bytes[16] = {};
    int dev = 0;
    FF_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16 || attr_bytes[dev].load(std::memory_order_acquire) < lds_bytes) {
      FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_general_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev >= 0 && dev < 16) attr_bytes[dev].store(160 * 1024, std::memory_order_release);
    }
  }
  FFProfScope prof(FF_CAT_ATTN, 4.0 * d.head_dim * (double)p.units * d.nk, st);
  const long long want = (p.units + GA_WAVES - 1) / GA_WAVES;
  const long long cap = (long long)ff_num_cus() * 16;
  const int blocks = (int)(want < cap ? want : cap);
  hipLaunchKernelGGL(attention_general_kernel, dim3(blocks), dim3(GA_WAVES * FF_WAVE), lds_bytes, st, p);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
