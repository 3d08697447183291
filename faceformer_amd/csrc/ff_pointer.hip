// Pointer head of the greedy decode: logits over the (variable-length) edge set of a wireframe,
// padding mask, arg-max with torch tie-breaking, feedback gather, stop-rule counters.
//
// One wavefront per sequence.  The projected decoder state p[b,:] lives in registers (E/64 floats
// per lane); edge-embedding rows of the wireframe are streamed with coalesced 16-byte loads (every
// sequence of a wireframe reads the same S x E block, so after the first wave it is served by L2),
// four rows in flight per iteration, and each logit is finished by a 64-lane butterfly reduction
// (__shfl_xor), which leaves the identical sum on every lane -- the comparison chain is therefore
// wave-uniform and needs no further communication.
#include <float.h>

#include "ff_common.h"
#include "ff_device.h"

namespace {


// one sequence by one wavefront; returns its token (wave-uniform)
template <int NV>
__device__ __forceinline__ int pointer_row(const PointerArgs& a, int b, int lane) {
  const int w = b / a.spg;
  const int nvec = a.E >> 2;
  f32x4 pv[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const int vi = lane + 64 * c;
    pv[c] = vi < nvec ? *reinterpret_cast<const f32x4*>(a.p + (size_t)b * a.ldp + vi * 4)
                      : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  int kv = a.S;
  if (a.kv_len) { const int k = a.kv_len[w]; kv = k < kv ? k : kv; }
  const float* mem = a.memory + (size_t)w * a.S * a.E;
  const unsigned char* mrow = a.mask ? a.mask + (size_t)w * a.S : nullptr;
  const unsigned char* erow = a.extra ? a.extra + (size_t)b * a.ldextra : nullptr;

  const float FILL = -FLT_MAX;  // torch.finfo(float32).min (reference utils.py:16-20)
  float best = -INFINITY, second = -INFINITY;
  int best_idx = 0;
  float keep = FILL;  // logit owned by this lane in the current block of 64 keys (trace output)

  for (int s0 = 0; s0 < kv; s0 += 4) {
    float part[4];
    bool live[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u;
      bool ok = s < kv;
      if (ok && mrow) ok = mrow[s] == 0;
      if (ok && erow) ok = erow[s] == 0;
      live[u] = ok;  // wave-uniform
      float acc = 0.f;
      if (ok) {
        const float* row = mem + (size_t)s * a.E;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          const int vi = lane + 64 * c;
          if (vi < nvec) {
            const f32x4 e = *reinterpret_cast<const f32x4*>(row + vi * 4);
            acc += (e.x * pv[c].x + e.y * pv[c].y) + (e.z * pv[c].z + e.w * pv[c].w);
          }
        }
      }
      part[u] = acc;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) part[u] += __shfl_xor(part[u], off, FF_WAVE);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int s = s0 + u;
      if (s >= kv) break;
      const float v = live[u] ? part[u] : FILL;
      if (v > best) { second = best; best = v; best_idx = s; }
      else if (v > second) { second = v; }
      if (a.logits) {
        if ((s & 63) == lane) keep = v;
        if ((s & 63) == 63) {
          a.logits[(size_t)b * a.ldlogits + (s - 63) + lane] = keep;
          keep = FILL;
        }
      }
    }
  }
  // keys >= kv are masked: they matter only for the trace and for the runner-up value
  if (kv < a.S) {
    if (FILL > best) { second = best; best = FILL; best_idx = kv; }
    else if (FILL > second) second = FILL;
    if (a.S - kv > 1 && FILL > second) second = FILL;
  }
  if (a.logits) {
    // flush the partially filled block, then the masked tail
    const int done = kv & ~63;
    if (kv & 63) {
      if (lane < (kv & 63)) a.logits[(size_t)b * a.ldlogits + done + lane] = keep;
    }
    for (int s = kv + lane; s < a.S; s += 64) a.logits[(size_t)b * a.ldlogits + s] = FILL;
  }
  if (best == -INFINITY) { best = FILL; best_idx = 0; }  // S == 0 cannot happen; defensive
  if (lane == 0) {
    a.next_tok[b] = best_idx;
    if (a.best) a.best[b] = best;
    if (a.second) a.second[b] = second;
  }
  if (a.next_rows) {
    const float* src = mem + (size_t)best_idx * a.E;
    float* dst = a.next_rows + (size_t)b * a.ldnext;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int vi = lane + 64 * c;
      if (vi < nvec) *reinterpret_cast<f32x4*>(dst + vi * 4) = *reinterpret_cast<const f32x4*>(src + vi * 4);
    }
  }
  return best_idx;
}

template <int NV>
__global__ __launch_bounds__(256) void pointer_kernel(PointerArgs a) {
  __shared__ int s_tok[4];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b < a.B) s_tok[threadIdx.x >> 6] = pointer_row<NV>(a, b, lane);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int b0 = blockIdx.x * (blockDim.x >> 6);
    ff_pointer_count_block(a, b0, s_tok, a.B - b0 < 4 ? a.B - b0 : 4);
  }
}

// GEMM path, stage 2 (ff_pointer_reduce_row, ff_device.h): mask the raw logit row in place and reduce it.
__global__ __launch_bounds__(256) void pointer_reduce_kernel(PointerArgs a) {
  __shared__ int s_tok[4];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b < a.B) s_tok[threadIdx.x >> 6] = ff_pointer_reduce_row(a, b, lane);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int b0 = blockIdx.x * (blockDim.x >> 6);
    ff_pointer_count_block(a, b0, s_tok, a.B - b0 < 4 ? a.B - b0 : 4);
  }
}

}  // namespace

extern "C" int ff_pointer_argmax(const float* p, int ldp, const float* memory, int S, int E,
                                 const unsigned char* mask, const int* kv_len,
                                 const unsigned char* extra_mask, int ldextra, int B,
                                 int seqs_per_group, int* next_tok, float* best, float* second,
                                 float* logits, int ldlogits, float* next_rows, int ldnext,
                                 int* count_ge, int ge_bound, int* count_eq, int eq_value,
                                 ff_stream_t stream) {
  return ff_pointer_argmax_sync(p, ldp, memory, S, E, mask, kv_len, extra_mask, ldextra, B, seqs_per_group, next_tok, best,
                                second, logits, ldlogits, next_rows, ldnext, count_ge, ge_bound, count_eq, eq_value, nullptr,
                                stream);
}

// The same operator with the decode engine's counter hand-over (ff_common.h: ff_pointer_sync); not part of the C ABI.
int ff_pointer_argmax_sync(const float* p, int ldp, const float* memory, int S, int E, const unsigned char* mask,
                           const int* kv_len, const unsigned char* extra_mask, int ldextra, int B, int seqs_per_group,
                           int* next_tok, float* best, float* second, float* logits, int ldlogits, float* next_rows,
                           int ldnext, int* count_ge, int ge_bound, int* count_eq, int eq_value,
                           const ff_pointer_sync* sync, ff_stream_t stream) {
  if (B == 0) return FF_OK;
  FF_CHECK_ARG(B > 0 && S > 0 && E > 0 && (E & 3) == 0 && E <= 2048 && seqs_per_group > 0,
               "ff_pointer_argmax: bad sizes B=%d S=%d E=%d", B, S, E);
  const bool logits_ready = sync && sync->logits_ready;
  FF_CHECK_ARG((p || logits_ready) && memory && next_tok, "ff_pointer_argmax: null pointer");
  FF_CHECK_ARG(!logits_ready || (logits && (B % seqs_per_group) == 0), "ff_pointer_argmax: logits_ready without a logits buffer");
  FF_CHECK_ARG(!(sync && sync->next_stats) || (logits && (B % seqs_per_group) == 0 && (E & 31) == 0),
               "ff_pointer_argmax: next_stats needs the GEMM path (a logits buffer) and E %% 32 == 0");
  FF_CHECK_ARG((logits_ready && !p) || ((ldp & 3) == 0 && ff_aligned16(p)), "ff_pointer_argmax: p misaligned");
  FF_CHECK_ARG(ff_aligned16(memory), "ff_pointer_argmax: memory misaligned");
  FF_CHECK_ARG(!next_rows || ((ldnext & 3) == 0 && ff_aligned16(next_rows)), "ff_pointer_argmax: next_rows misaligned");
  FF_CHECK_ARG(!logits || ldlogits >= S, "ff_pointer_argmax: ldlogits < S");
  FF_CHECK_ARG(!extra_mask || ldextra >= S, "ff_pointer_argmax: ldextra < S");
  PointerArgs a{p, ldp, memory, S, E, mask, kv_len, extra_mask, ldextra, B, seqs_per_group,
                next_tok, best, second, logits, ldlogits, next_rows, ldnext,
                count_ge, ge_bound, count_eq, eq_value,
                sync ? sync->seen : nullptr, sync ? sync->arrive : nullptr, sync ? sync->host_slot : nullptr,
                sync ? sync->host_which : 0, sync ? sync->next_stats : nullptr};
  FF_CHECK_ARG(!a.arrive || (a.host_slot && (a.host_which ? count_eq : count_ge)), "ff_pointer_argmax: counter hand-over without a counter");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ff_cdiv(B, 4)), block(256);
  if (logits != nullptr && (B % seqs_per_group) == 0) {
    // logits[w*spg + f, s] = < p[w*spg + f, :], memory[w, s, :] > : one GEMM problem per wireframe
    if (!logits_ready)
      FF_RETURN_IF(ff_gemm_f32_batched(p, ldp, nullptr, 0, memory, E, nullptr, nullptr, 0, logits, ldlogits,
                                       seqs_per_group, S, E, 0, 0, B / seqs_per_group,
                                       (long long)seqs_per_group * ldp, (long long)S * E,
                                       (long long)seqs_per_group * ldlogits, stream));
    FFProfScope prof(FF_CAT_POINTER, (double)B * S * 8.0, st);
    hipLaunchKernelGGL(pointer_reduce_kernel, grid, block, 0, st, a);
    FF_CHECK_LAUNCH();
    return FF_OK;
  }
  FFProfScope prof(FF_CAT_POINTER, 2.0 * B * (double)S * E, st);
  const int nv = ff_cdiv(E / 4, 64);
  if (nv <= 1) hipLaunchKernelGGL(pointer_kernel<1>, grid, block, 0, st, a);
  else if (nv <= 2) hipLaunchKernelGGL(pointer_kernel<2>, grid, block, 0, st, a);
  else if (nv <= 4) hipLaunchKernelGGL(pointer_kernel<4>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(pointer_kernel<8>, grid, block, 0, st, a);
  FF_CHECK_LAUNCH();
  return FF_OK;
}
