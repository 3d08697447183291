"""ctypes binding of libfaceformer_hip.so -- the C ABI declared in include/faceformer_hip.h.

There is deliberately no fallback: if the shared library is missing or fails to load, `load()` raises
and every op / model of this package fails loudly (the product path must never run on a substitute).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FF_HIP_LIB") or os.path.join(HERE, "libfaceformer_hip.so")   # (override: A/B of kernel builds, tools/)

FF_MAX_LAYERS = 16
FF_HEAD_DIM = 64
FF_PARALLEL, FF_SEQ2SEQ = 0, 1
FF_REUSE_LAYER0_QKV, FF_LAST_LAYER_LAST_ROW, FF_RETURN_POINTER, FF_NO_STOP, FF_DEDUP_PAD_ANCHORS = 1, 2, 4, 8, 16
FF_FUSE_LAYERNORM = 32
FF_STOP_EACH_EOS = 512
FF_NO_L0_FOLD = 1024
FF_NO_POINTER_FOLD = 2048
FF_ABI_VERSION = 103   # include/faceformer_hip.h: the struct layouts below are this version's

fptr = C.c_void_p  # device pointers travel as integers


class HipExtensionError(RuntimeError):
    pass


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", fptr), ("k", fptr), ("v", fptr), ("o", fptr),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
        ("num_groups", C.c_int), ("num_heads", C.c_int),
        ("nq", C.c_int),
        ("q_group_stride", C.c_int), ("q_inner", C.c_int), ("q_outer_stride", C.c_int),
        ("nk", C.c_int),
        ("k_group_stride", C.c_int), ("k_stride", C.c_int),
        ("kv_len", fptr),
        ("key_mask", fptr), ("mask_stride", C.c_int),
        ("causal", C.c_int),
        ("scale", C.c_float),
        ("kv_planes", fptr),
    ]


class AttnGeneralDesc(C.Structure):
    _fields_ = [
        ("q", fptr), ("k", fptr), ("v", fptr), ("o", fptr),
        ("ldq", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("ldo", C.c_int),
        ("num_groups", C.c_int), ("num_heads", C.c_int), ("head_dim", C.c_int),
        ("nq", C.c_int),
        ("q_group_stride", C.c_int), ("q_inner", C.c_int), ("q_outer_stride", C.c_int),
        ("nk", C.c_int),
        ("k_group_stride", C.c_int), ("k_stride", C.c_int),
        ("kv_len", fptr),
        ("key_mask", fptr), ("mask_stride", C.c_int),
        ("causal", C.c_int),
        ("attn_bias", fptr),
        ("attn_mask", fptr),
        ("attn_ld", C.c_int),
        ("attn_batch_stride", C.c_longlong),
        ("scale", C.c_float),
    ]


class GemmLnDesc(C.Structure):
    _fields_ = [
        ("A", fptr), ("lda", C.c_int), ("W", fptr), ("ldw", C.c_int), ("bias", fptr),
        ("residual", fptr), ("ldr", C.c_int), ("C", fptr), ("ldc", C.c_int),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("act", C.c_int), ("tile", C.c_int),
        ("ln_stats_in", fptr), ("ln_nseg", C.c_int), ("ln_eps", C.c_float),
        ("row_table", fptr), ("ld_row_table", C.c_int), ("row_div", C.c_int), ("row_cols", C.c_int),
        ("ln_stats_out", fptr),
    ]


class MhaWeights(C.Structure):
    _fields_ = [("in_proj_w", fptr), ("in_proj_b", fptr), ("out_w", fptr), ("out_b", fptr)]


class LayerWeights(C.Structure):
    _fields_ = [
        ("self_attn", MhaWeights), ("cross_attn", MhaWeights),
        ("lin1_w", fptr), ("lin1_b", fptr), ("lin2_w", fptr), ("lin2_b", fptr),
        ("norm1_w", fptr), ("norm1_b", fptr), ("norm2_w", fptr), ("norm2_b", fptr),
        ("norm3_w", fptr), ("norm3_b", fptr),
        ("in_proj_planes", fptr), ("lin1_planes", fptr), ("lin2_planes", fptr),
        ("self_out_planes", fptr), ("cross_q_planes", fptr), ("cross_out_planes", fptr),
        ("ln1_w", fptr), ("ln1_b", fptr), ("ln1_pos", fptr),
        ("ln2_w", fptr), ("ln2_b", fptr), ("ln2_pos", fptr),
        ("ln3_w", fptr), ("ln3_b", fptr),
        ("ln1_planes", fptr), ("ln2_planes", fptr), ("ln3_planes", fptr),
        ("ln1_csum", fptr), ("ln2_csum", fptr), ("ln3_csum", fptr),
    ]


class Model(C.Structure):
    _fields_ = [
        ("E", C.c_int), ("H", C.c_int), ("FF", C.c_int),
        ("num_enc_layers", C.c_int), ("num_dec_layers", C.c_int),
        ("in_dim", C.c_int), ("num_token", C.c_int),
        ("pos_len", C.c_int), ("qpos_len", C.c_int),
        ("ln_eps", C.c_float),
        ("tok_embed", fptr),
        ("emb_w1", fptr), ("emb_b1", fptr), ("emb_w2", fptr), ("emb_b2", fptr),
        ("pos_table", fptr), ("qpos_table", fptr),
        ("enc", LayerWeights * FF_MAX_LAYERS),
        ("enc_norm_w", fptr), ("enc_norm_b", fptr),
        ("dec", LayerWeights * FF_MAX_LAYERS),
        ("dec_norm_w", fptr), ("dec_norm_b", fptr),
        ("proj_w", fptr), ("proj_b", fptr),
        ("proj_fold_w", fptr), ("proj_fold_b", fptr),
        ("split_kind", C.c_int),
    ]


class DecodeParams(C.Structure):
    _fields_ = [
        ("variant", C.c_int), ("N", C.c_int), ("L", C.c_int), ("F", C.c_int), ("T", C.c_int),
        ("chunk_wireframes", C.c_int), ("chunk_seqs", C.c_int), ("num_streams", C.c_int),
        ("sync_every", C.c_int), ("flags", C.c_int),
        ("tok_sos", C.c_int), ("tok_eos", C.c_int),
        ("x3_min_rows", C.c_int), ("chunk_max_seqs", C.c_int), ("ln_fuse_max_rows", C.c_int),
        ("stop_fn", C.c_void_p), ("stop_user", C.c_void_p),
    ]


# ff_stop_fn: int (*)(void* user, const int* step_counts, int num_steps)
STOP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_int)


# name -> (restype, argtypes); must list every symbol of include/faceformer_hip.h
SIGNATURES = {
    "ff_version": (C.c_int, []),
    "ff_last_error": (C.c_char_p, []),
    "ff_device_count": (C.c_int, []),
    "ff_profile_begin": (C.c_int, []),
    "ff_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.c_int]),
    "ff_profile_bytes": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "ff_profile_bracket_us": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "ff_clock_probe_launch": (C.c_int, [C.c_double, C.c_void_p]),
    "ff_clock_probe_read": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "ff_layernorm": (C.c_int, [fptr, C.c_int, fptr, fptr, C.c_float, fptr, C.c_int, fptr, C.c_int,
                               fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "ff_add_pos": (C.c_int, [fptr, C.c_int, fptr, C.c_int, C.c_int, C.c_int, fptr, C.c_int, C.c_int,
                             C.c_int, fptr]),
    "ff_gelu": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr]),
    "ff_gemm_f32": (C.c_int, [fptr, C.c_int, fptr, C.c_int, fptr, C.c_int, fptr, fptr, C.c_int, fptr,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "ff_gemm_f32_batched": (C.c_int, [fptr, C.c_int, fptr, C.c_int, fptr, C.c_int, fptr, fptr, C.c_int, fptr,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_longlong, C.c_longlong, C.c_longlong, fptr]),
    "ff_gemm_f32_ln": (C.c_int, [C.POINTER(GemmLnDesc), fptr]),
    "ff_fold_layernorm_linear": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr, fptr, fptr, fptr, C.c_int, C.c_int,
                                           C.c_int, fptr, fptr, fptr, fptr]),
    "ff_split_weight_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ff_split_weight_bf16x3": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr, fptr]),
    "ff_gemm_x3": (C.c_int, [fptr, C.c_int, fptr, C.c_int, fptr, fptr, fptr, C.c_int, fptr, C.c_int,
                             C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "ff_gemm_x3_ln": (C.c_int, [C.POINTER(GemmLnDesc), fptr, C.c_int, C.c_int, fptr, fptr]),
    "ff_split_weight_fp16x2_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ff_split_weight_fp16x2": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr, fptr]),
    "ff_gemm_x2h": (C.c_int, [fptr, C.c_int, fptr, C.c_int, fptr, fptr, fptr, C.c_int, fptr, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, fptr]),
    "ff_gemm_x2h_ln": (C.c_int, [C.POINTER(GemmLnDesc), fptr, C.c_int, C.c_int, fptr, fptr]),
    "ff_set_x3_tuning": (C.c_int, [C.c_int]),
    "ff_attention": (C.c_int, [C.POINTER(AttnDesc), fptr]),
    "ff_attention_general": (C.c_int, [C.POINTER(AttnGeneralDesc), fptr]),
    "ff_set_attention_algo": (C.c_int, [C.c_int]),
    "ff_attention_planes_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ff_attention_split_kv": (C.c_int, [fptr, fptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fptr, fptr]),
    "ff_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "ff_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "ff_reset_tuning": (C.c_int, []),
    "ff_set_gemm_tuning": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ff_pointer_argmax": (C.c_int, [fptr, C.c_int, fptr, C.c_int, C.c_int, fptr, fptr, fptr, C.c_int,
                                    C.c_int, C.c_int, fptr, fptr, fptr, fptr, C.c_int, fptr, C.c_int,
                                    fptr, C.c_int, fptr, C.c_int, fptr]),
    "ff_gather_rows": (C.c_int, [fptr, C.c_int, C.c_int, fptr, C.c_int, C.c_int, fptr, C.c_int, fptr]),
    "ff_assemble_embedding": (C.c_int, [fptr, C.c_int, fptr, C.c_int, C.c_int, C.c_int, C.c_int, fptr,
                                        fptr]),
    "ff_prepare_mask": (C.c_int, [fptr, C.c_int, C.c_int, C.c_int, fptr, fptr, fptr]),
    "ff_encode_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.c_int, C.c_int]),
    "ff_encode": (C.c_int, [C.POINTER(Model), fptr, fptr, fptr, C.c_int, C.c_int, fptr, fptr,
                            C.c_size_t, fptr]),
    "ff_decode_workspace_bytes": (C.c_size_t, [C.POINTER(Model), C.POINTER(DecodeParams), C.POINTER(C.c_int)]),
    "ff_decode": (C.c_int, [C.POINTER(Model), C.POINTER(DecodeParams), fptr, fptr, fptr, fptr,
                            C.POINTER(C.c_int), fptr,
                            fptr, C.POINTER(C.c_int), C.POINTER(C.c_int), fptr, fptr, fptr, fptr, fptr, fptr,
                            C.c_size_t, fptr]),
    "ff_gemm_prepare_stream": (C.c_int, [fptr]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises HipExtensionError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters inside a torch process: torch ships its own libamdhip64 (same SONAME as the
    # system ROCm one this library is linked against).  Importing torch first makes the dynamic loader
    # resolve our HIP calls to the runtime torch already initialised -- two runtimes in one process
    # cannot see each other's devices or streams.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            "HIP extension not built: %s is missing. Run `python -m faceformer_amd.hip.build` "
            "(needs hipcc); there is no CPU fallback for the decode path." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipExtensionError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipExtensionError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.ff_version() != FF_ABI_VERSION:   # a stale .so would read the ctypes structs above with another layout
        raise HipExtensionError("%s reports ABI version %d, this package binds version %d: rebuild it "
                                "(python -m faceformer_amd.hip.build --force)" % (LIB_PATH, lib.ff_version(), FF_ABI_VERSION))
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ff_last_error()
        raise HipExtensionError("%s failed with status %d: %s" % (what, status, (msg or b"").decode()))
