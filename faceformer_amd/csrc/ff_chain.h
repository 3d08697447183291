// Chain kernel (ff_chain.hip): operator descriptors and the host-side recorder.  While a recording is active on the calling
// thread, ff_gemm_f32* / ff_gemm_f32_ln / ff_attention / ff_layernorm / ff_pointer_argmax APPEND their operator to the chain
// instead of launching it; ff_chain_end() then enqueues ONE persistent launch that runs the recorded operators in order.
#pragma once
#include "ff_common.h"
#include "ff_device.h"

enum { FF_CH_GEMM = 1, FF_CH_ATTN = 2, FF_CH_LN = 3, FF_CH_PTR = 4, FF_CH_GEMM64 = 5 };
constexpr int FF_FLOW_MAX_OPS = 4;

struct ff_chain_op {
  int kind;
  int barrier;        // 1: a grid-wide phase boundary follows this operator (0: the next one does not depend on it)
  int units;          // work items dealt round-robin to the workgroups: 32x32 tiles / attention blocks / rows / sequences
  int aux0, aux1;     // attention: query tiles per (group, head), key split; LayerNorm: float4 chunks per lane
  long total_units;   // attention: (group, head, query tile) units
  union {
    GemmArgs g;
    ff_attn_desc a;
    LnArgs ln;
    PointerArgs p;
  } u;
};

#ifdef FF_EXPERIMENTAL
bool ff_chain_recording();
void ff_chain_next_is_independent();   // hint: the next recorded operator may run in the same phase as the previous one
bool ff_chain_gemm_ok(const GemmArgs& g, int batch);
int ff_chain_record_gemm(const GemmArgs& g, int batch);
int ff_chain_record_attention(const ff_attn_desc& d);
int ff_chain_record_layernorm(const LnArgs& a);
int ff_chain_record_pointer(const PointerArgs& a);
int ff_chain_prepare(size_t ops_needed, hipStream_t st);
int ff_chain_begin();
void ff_chain_abort();
int ff_chain_end(hipStream_t st, int* launched);
int ff_chain_check(hipStream_t st);
// Flow launches (ff_gemm.hip: gemm_flow_kernel): between ff_flow_begin() and ff_flow_end() every projection is recorded as a
// 64x64-tile operator that depends, row panel by row panel, on the projection recorded before it.
int ff_flow_begin();
int ff_flow_end(hipStream_t st, int* launched);
bool ff_flow_recording();
int ff_gemm_flow_launch(const ff_chain_op* dev_ops, int nops, unsigned* ctr, unsigned* done, unsigned* err, int panel_stride,
                        hipStream_t st);
#else
// Default build: the persistent-launch experiments (chain / flow launches, step graphs) are compiled OUT -- built, parity-tested
// and measured slower than launch-per-operator in round 3.  `python -m faceformer_amd.hip.build --experimental`
// builds libfaceformer_hip_exp.so with them; ff_decode of the default library refuses FF_CHAIN / FF_FLOW / FF_GRAPH.
inline bool ff_chain_recording() { return false; }
inline void ff_chain_next_is_independent() {}
inline bool ff_chain_gemm_ok(const GemmArgs&, int) { return false; }
inline int ff_chain_record_gemm(const GemmArgs&, int) { return FF_ERR_ARG; }
inline int ff_chain_record_attention(const ff_attn_desc&) { return FF_ERR_ARG; }
inline int ff_chain_record_layernorm(const LnArgs&) { return FF_ERR_ARG; }
inline int ff_chain_record_pointer(const PointerArgs&) { return FF_ERR_ARG; }
inline int ff_chain_prepare(size_t, hipStream_t) { return FF_OK; }
inline int ff_chain_begin() { return FF_ERR_ARG; }
inline void ff_chain_abort() {}
inline int ff_chain_end(hipStream_t, int* launched) { if (launched) *launched = 0; return FF_OK; }
inline int ff_chain_check(hipStream_t) { return FF_OK; }
inline int ff_flow_begin() { return FF_ERR_ARG; }
inline int ff_flow_end(hipStream_t, int* launched) { if (launched) *launched = 0; return FF_OK; }
inline bool ff_flow_recording() { return false; }
#endif
