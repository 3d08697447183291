// Cross-attention on the fp16 matrix cores with fp32 accuracy ("2 x fp16", round 6): softmax(q k^T * scale + mask) v for key sets of at
// most 288 rows shared by many query tiles -- the decoder cross-attention of the 64..256-edge configurations (reference
// transformer.py:248-251) -- with the products evaluated like ff_gemm_x2h's: every operand x = x1 + x2, x1 = fp16(x),
// x2' = fp16((x - x1) 2^11), three products x1 y1 + (x1 y2' + x2' y1) 2^-11 on v_mfma_f32_32x32x16_f16 (x1 y1 in its own
// accumulator).  An item (32 queries x 32 keys) is 24 MFMAs of 32 cycles instead of 65 of 64 on the f32 matrix cores.
//
// K and V of a (group, head) pair are constant for a whole decode (cross-attention keys are the encoder memory): they are split
// ONCE per batch into fp16 planes (ff_attention_split_kv) laid out as the kernel's LDS image --
//     K1 | K2' : [288 keys][64] fp16, the eight 16-byte chunks of a row XOR-swizzled with (key >> 1) & 7 (ds_read_b128 fragments
//                of 32 consecutive keys then touch 16 different bank slots per lane group),
//     V1t | V2t' : [64 d][288 keys (+ 4 pad)] fp16, TRANSPOSED (the P.V product contracts over keys: an operand lane holds
//                consecutive keys of one d), row stride 584 bytes (ds_read_b64 of 32 consecutive d: conflict free),
// 148 480 bytes per pair -- and reach LDS by one linear LDS-DMA copy per block.  Queries and the softmax weights are split in
// registers.  Work: the pair's query tiles are dealt round-robin to the 8 waves of its c blocks (whole tiles: no partial records).
//
// Range: fp16 has five exponent bits.  q, k, v are projections of LayerNorm output (bounded by sqrt(E) ||w_n||_2 + |b_n|; the engine
// checks the bounds when it binds the planes), the weights are in [0, 1].  Rows past a group's kv_len are zero in the planes and
// masked by the additive bias.
#include <atomic>

#include "ff_common.h"
#include "ff_device.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int XK_KEYS = 288;
constexpr int XK_KROW = 128;                       // bytes of a K row (64 fp16)
constexpr int XK_VROW = 584;                       // bytes of a V^T row (288 + 4 fp16)
constexpr int XK_K_BYTES = XK_KEYS * XK_KROW;      // 36 864
constexpr int XK_V_BYTES = 64 * XK_VROW;           // 37 376
constexpr int XK_PLANE_BYTES = 2 * XK_K_BYTES + 2 * XK_V_BYTES;   // 148 480 = 145 KB
constexpr int XK_LDS_BYTES = XK_PLANE_BYTES + XK_KEYS * 4;
constexpr int XK_NW = 8;
static_assert(XK_PLANE_BYTES % 1024 == 0, "the planes are copied in 1 KB wave-instructions");

// x (two floats) -> packed fp16 pairs of the two terms (second term at 2^11)
__device__ __forceinline__ void split_h2(float x0, float x1, unsigned& p1, unsigned& p2) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f16x2 h = __builtin_convertvector(f32x2{x0, x1}, f16x2);
  p1 = __builtin_bit_cast(unsigned, h);
  const f16x2 l = __builtin_convertvector(f32x2{(x0 - (float)h[0]) * 2048.0f, (x1 - (float)h[1]) * 2048.0f}, f16x2);
  p2 = __builtin_bit_cast(unsigned, l);
}

// ---- K | V of every (group, head) pair -> the planes (once per batch and layer) ----------------------------------------------------
__global__ __launch_bounds__(256) void kv_split_kernel(const float* __restrict__ k, const float* __restrict__ v, int ldk, int ldv,
                                                       int num_heads, int nk, int k_group_stride, int k_stride,
                                                       unsigned char* __restrict__ planes, long long plane_stride) {
  const int pair = blockIdx.y, g = pair / num_heads, h = pair % num_heads;
  unsigned char* out = planes + (size_t)pair * plane_stride;
  const float* kb = k + (size_t)g * k_group_stride * ldk + h * FF_HEAD_DIM;
  const float* vb = v + (size_t)g * k_group_stride * ldv + h * FF_HEAD_DIM;
  // one thread: key pair (key, key + 1) x d pair (d, d + 1)?  Simpler and coalesced enough for a once-per-batch kernel:
  // thread = (key, d pair): K rows are written as fp16 pairs along d, V^T as single halfs along keys
  const int total = XK_KEYS * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int key = i >> 5, d = (i & 31) * 2;
    float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
    if (key < nk) {
      const size_t row = (size_t)key * k_stride;
      k0 = kb[row * ldk + d]; k1 = kb[row * ldk + d + 1];
      v0 = vb[row * ldv + d]; v1 = vb[row * ldv + d + 1];
    }
    unsigned a1, a2, b1, b2;
    split_h2(k0, k1, a1, a2);
    split_h2(v0, v1, b1, b2);
    const int chunk = (d >> 3) ^ ((key >> 1) & 7);
    const size_t koff = (size_t)key * XK_KROW + chunk * 16 + (d & 7) * 2;
    *reinterpret_cast<unsigned*>(out + koff) = a1;
    *reinterpret_cast<unsigned*>(out + XK_K_BYTES + koff) = a2;
    unsigned short* vt1 = reinterpret_cast<unsigned short*>(out + 2 * XK_K_BYTES);
    unsigned short* vt2 = reinterpret_cast<unsigned short*>(out + 2 * XK_K_BYTES + XK_V_BYTES);
    vt1[(size_t)d * (XK_VROW / 2) + key] = (unsigned short)(b1 & 0xffffu);
    vt1[(size_t)(d + 1) * (XK_VROW / 2) + key] = (unsigned short)(b1 >> 16);
    vt2[(size_t)d * (XK_VROW / 2) + key] = (unsigned short)(b2 & 0xffffu);
    vt2[(size_t)(d + 1) * (XK_VROW / 2) + key] = (unsigned short)(b2 >> 16);
  }
  // the four pad keys of every V^T row (never read as data; keep them defined)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 64 * 4; i += gridDim.x * blockDim.x) {
    const int d = i >> 2, key = XK_KEYS + (i & 3);
    reinterpret_cast<unsigned short*>(out + 2 * XK_K_BYTES)[(size_t)d * (XK_VROW / 2) + key] = 0;
    reinterpret_cast<unsigned short*>(out + 2 * XK_K_BYTES + XK_V_BYTES)[(size_t)d * (XK_VROW / 2) + key] = 0;
  }
}

__device__ __forceinline__ u32x4 xk_read16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ u32x2 xk_read8(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

__global__ __launch_bounds__(64 * XK_NW, 1) void attention_x2h_kernel(ff_attn_desc d, const unsigned char* __restrict__ planes,
                                                                       long long plane_stride, int P, int c, int q_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const Ms = reinterpret_cast<float*>(lds + XK_PLANE_BYTES);   // additive key bias: 0 or -inf
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
  const float qscale = d.scale * 1.4426950408889634f;
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
  const int nblk = gridDim.x;
  const int rank = (P <= nblk) ? blockIdx.x / P : 0;
  for (int pair = (P <= nblk) ? blockIdx.x % P : blockIdx.x; pair < P; pair += (P <= nblk ? P : nblk)) {
    const int g = pair / d.num_heads, h = pair % d.num_heads;
    int nk = d.nk;
    if (d.kv_len) { const int kl = d.kv_len[g]; nk = kl < nk ? kl : nk; }
    unsigned char mbyte = 0;
    if (d.key_mask && tid < XK_KEYS && tid < d.nk) mbyte = d.key_mask[(size_t)g * d.mask_stride + tid];
    {   // the pair's planes: one linear copy, 1 KB per wave-instruction
      const unsigned char* src = planes + (size_t)pair * plane_stride;
      for (int q = wave; q < XK_PLANE_BYTES / 1024; q += XK_NW)
        __builtin_amdgcn_global_load_lds(src + (size_t)q * 1024 + lane * 16,
                                         (__attribute__((address_space(3))) void*)(lds + q * 1024), 16, 0, 0);
    }
    if (tid < XK_KEYS) Ms[tid] = (tid >= nk || mbyte != 0) ? -INFINITY : 0.f;
    __syncthreads();   // (waits for the LDS-DMA: it counts in vmcnt)
    const int ntiles = (nk + 31) >> 5;
    const int slot = rank * XK_NW + wave, nslots = c * XK_NW;
    for (int qt = slot; qt < q_tiles; qt += nslots) {
      // ---- this wave's query tile: lane (query l32, half) holds d = 16 ks + 8 half + 0..7 for ks = 0..3, scaled and split ----
      const int qi = qt * 32 + l32;
      const bool qv = qi < d.nq;
      const int qc = qv ? qi : d.nq - 1;
      const size_t qrow = (size_t)g * d.q_group_stride + (size_t)(qc / d.q_inner) * d.q_outer_stride + (size_t)(qc % d.q_inner);
      u32x4 q1[4], q2[4];
      {
        const float* qp = d.q + qrow * d.ldq + h * FF_HEAD_DIM + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 16 * ks);
          const f32x4 b = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4);
          unsigned p1[4], p2[4];
          split_h2(a.x * qscale, a.y * qscale, p1[0], p2[0]);
          split_h2(a.z * qscale, a.w * qscale, p1[1], p2[1]);
          split_h2(b.x * qscale, b.y * qscale, p1[2], p2[2]);
          split_h2(b.z * qscale, b.w * qscale, p1[3], p2[3]);
          q1[ks] = u32x4{p1[0], p1[1], p1[2], p1[3]};
          q2[ks] = u32x4{p2[0], p2[1], p2[2], p2[3]};
        }
      }
      float m_run = -INFINITY, l_run = 0.f;
      f32x16 om[2], os[2];   // O^T accumulators (d tile 0 / 1): x1 y1 products / the two small products (at 2^11)
#pragma unroll
      for (int e = 0; e < 16; ++e) { om[0][e] = 0.f; om[1][e] = 0.f; os[0][e] = 0.f; os[1][e] = 0.f; }
      for (int kt = 0; kt < ntiles; ++kt) {
        // ---- S^T tile: 32 keys x 32 queries ----
        f32x16 sm, ss;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sm[e] = 0.f; ss[e] = 0.f; }
        const int key = kt * 32 + l32;
        const unsigned ka = lds0 + key * XK_KROW;
        const unsigned sw = (unsigned)((key >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const unsigned off = (((unsigned)(2 * ks + half)) ^ sw) << 4;
          u32x4 k1 = xk_read16(ka + off);
          u32x4 k2 = xk_read16(ka + XK_K_BYTES + off);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k1), "+v"(k2)::"memory");   // (ties the fragments to the wait: the MFMAs stay behind it)
          sm = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k1), __builtin_bit_cast(f16x8, q1[ks]), sm, 0, 0, 0);
          ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k1), __builtin_bit_cast(f16x8, q2[ks]), ss, 0, 0, 0);
          ss = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, k2), __builtin_bit_cast(f16x8, q1[ks]), ss, 0, 0, 0);
        }
        // ---- online softmax over this tile's keys (register e = 4 m + r  <->  key 8 m + 4 half + r of the tile) ----
        float s[16];
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float bias = Ms[kt * 32 + 8 * (e >> 2) + 4 * half + (e & 3)];
          s[e] = sm[e] + ss[e] * (1.0f / 2048.0f) + bias;
          tmax = fmaxf(tmax, s[e]);
        }
        tmax = ff_halves_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = ff_exp2(m_run - m_safe);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = ff_exp2(s[e] - m_safe); psum += s[e]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (!__all(alpha == 1.0f)) {
#pragma unroll
          for (int e = 0; e < 16; ++e) { om[0][e] *= alpha; om[1][e] *= alpha; os[0][e] *= alpha; os[1][e] *= alpha; }
        }
        // ---- weights -> fp16 terms: step st takes registers 8 st .. 8 st + 7 (keys 16 st + 4 half + r and 16 st + 8 + 4 half + r) ----
        u32x4 p1[2], p2[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          unsigned a1[4], a2[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) split_h2(s[8 * st + 2 * j], s[8 * st + 2 * j + 1], a1[j], a2[j]);
          p1[st] = u32x4{a1[0], a1[1], a1[2], a1[3]};
          p2[st] = u32x4{a2[0], a2[1], a2[2], a2[3]};
        }
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned va = lds0 + 2 * XK_K_BYTES + (dt * 32 + l32) * XK_VROW + (kt * 32 + 4 * half) * 2;
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            u32x2 v1a = xk_read8(va + 32 * st), v1b = xk_read8(va + 32 * st + 16);
            u32x2 v2a = xk_read8(va + XK_V_BYTES + 32 * st), v2b = xk_read8(va + XK_V_BYTES + 32 * st + 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v1a), "+v"(v1b), "+v"(v2a), "+v"(v2b)::"memory");
            const u32x4 v1 = u32x4{v1a.x, v1a.y, v1b.x, v1b.y}, v2 = u32x4{v2a.x, v2a.y, v2b.x, v2b.y};
            om[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, v1), __builtin_bit_cast(f16x8, p1[st]), om[dt], 0, 0, 0);
            os[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, v1), __builtin_bit_cast(f16x8, p2[st]), os[dt], 0, 0, 0);
            os[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, v2), __builtin_bit_cast(f16x8, p1[st]), os[dt], 0, 0, 0);
          }
        }
      }
      // ---- normalise and store: lane (query l32, half) holds d = dt * 32 + 8 m + 4 half + r ----
      const float l_tot = ff_halves_sum(l_run);
      const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
      if (qv) {
        float* op = d.o + qrow * d.ldo + h * FF_HEAD_DIM + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = (om[dt][g4 * 4 + r] + os[dt][g4 * 4 + r] * (1.0f / 2048.0f)) * inv;
            ff_st16(op + dt * 32 + 8 * g4, a);
          }
      }
    }
    __syncthreads();   // the next pair overwrites the planes
  }
}

}  // namespace

size_t ff_attention_planes_stride() { return (size_t)XK_PLANE_BYTES; }
bool ff_attention_x2h_ok(const ff_attn_desc& d) { return d.nk > 0 && d.nk <= XK_KEYS && !d.causal; }

// (internal: ff_attention hands eligible launches with planes over)
int ff_attention_x2h_launch(const ff_attn_desc& d, const void* planes, long long plane_stride, hipStream_t st) {
  static std::atomic<bool> attr_done[16] = {};
  int dev = 0;
  FF_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_done[dev].load(std::memory_order_acquire)) {
    FF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_x2h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     XK_LDS_BYTES));
    if (dev >= 0 && dev < 16) attr_done[dev].store(true, std::memory_order_release);
  }
  const long gh = (long)d.num_groups * d.num_heads;
  FF_CHECK_ARG(gh < 2147483647L, "ff_attention: too many (group, head) pairs");
  const int P = (int)gh, cus = ff_num_cus(), qt32 = ff_cdiv(d.nq, 32);
  int c = P <= cus ? cus / P : 1;
  const int cmax = ff_cdiv(qt32, XK_NW);   // blocks beyond one query tile per wave are idle
  if (c > cmax) c = cmax;
  if (c < 1) c = 1;
  const int nblocks = P <= cus ? P * c : cus;
  hipLaunchKernelGGL(attention_x2h_kernel, dim3(nblocks), dim3(64 * XK_NW), XK_LDS_BYTES, st, d,
                     static_cast<const unsigned char*>(planes), plane_stride, P, c, qt32);
  FF_CHECK_LAUNCH();
  return FF_OK;
}

extern "C" size_t ff_attention_planes_bytes(int num_groups, int num_heads) {
  if (num_groups <= 0 || num_heads <= 0) return 0;
  return (size_t)num_groups * num_heads * XK_PLANE_BYTES;
}

extern "C" int ff_attention_split_kv(const float* k, const float* v, int ldk, int ldv, int num_groups, int num_heads, int nk,
                                     int k_group_stride, int k_stride, void* planes, ff_stream_t stream) {
  FF_CHECK_ARG(k && v && planes && num_groups > 0 && num_heads > 0 && nk > 0 && nk <= XK_KEYS,
               "ff_attention_split_kv: bad arguments (1 <= nk <= %d)", XK_KEYS);
  FF_CHECK_ARG(ldk >= num_heads * FF_HEAD_DIM && ldv >= num_heads * FF_HEAD_DIM && ff_aligned16(planes),
               "ff_attention_split_kv: ld smaller than num_heads * 64, or planes not 16-byte aligned");
  FF_CHECK_ARG((long)num_groups * num_heads <= 65535, "ff_attention_split_kv: at most 65535 (group, head) pairs per call");
  hipLaunchKernelGGL(kv_split_kernel, dim3(9, num_groups * num_heads), dim3(256), 0, (hipStream_t)stream, k, v, ldk, ldv, num_heads, nk,
                     k_group_stride, k_stride, static_cast<unsigned char*>(planes), (long long)XK_PLANE_BYTES);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
