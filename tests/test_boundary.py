"""CPU: the drop-in boundary -- config-file API, module surface, state_dict layout, C-ABI exports,
and the no-fallback rule (the product path refuses to run without a ROCm device)."""
import argparse
import ctypes
import glob
import os
import pickle
import re

import pytest
import torch

from conftest import ROOT, token_ns


# ---- config-file API (reference faceformer/config.py:54-79) -------------------------------------------
def test_default_tree_and_yaml_overlay():
    from faceformer_amd.config import get_cfg, get_parser
    args = get_parser().parse_args(["--config-file", os.path.join(ROOT, "configs", "ours.yml"),
                                    "model.num_lines", "256", "trainer.lr", "0.5"])
    cfg = get_cfg(args)
    assert cfg.model_class == "SurfaceFormer_Parallel" and cfg.dataset_class == "ABCDataset_Parallel"
    assert cfg.model.num_lines == 256 and cfg.model.max_face_length == 37
    assert cfg.model.num_model == 512 and cfg.model.num_head == 8 and cfg.model.num_feedforward == 1024
    assert cfg.model.token.len == 4 and cfg.model.token.SOS == 1 and cfg.model.token.EOS == 3
    assert cfg.trainer.lr == 0.5 and cfg.post_process.enclosedness_tol == 2e-4
    assert cfg.is_frozen()
    with pytest.raises(AttributeError):
        cfg.model.num_lines = 1
    c2 = pickle.loads(pickle.dumps(cfg))
    assert c2.model.num_lines == 256 and c2.is_frozen()
    assert dict(**cfg.model)["num_lines"] == 256  # splatted into the model ctor by the harness


def test_every_reference_config_loads():
    from faceformer_amd.config import load_cfg
    want = {"ours.yml": ("SurfaceFormer_Parallel", 216, 37, None),
            "ours-perspective.yml": ("SurfaceFormer_Parallel", 202, 38, None),
            "ours-fixed_viewpoint.yml": ("SurfaceFormer_Parallel", 186, 33, None),
            "seq2seq.yml": ("SurfaceFormer", 110, None, 259),
            "seq2seq+coedge.yml": ("SurfaceFormer", 216, None, 259)}
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yml")))
    assert {os.path.basename(f) for f in files} == set(want)
    for f in files:
        cfg = load_cfg(f)
        cls, lines, tface, tseq = want[os.path.basename(f)]
        assert cfg.model_class == cls and cfg.model.num_lines == lines
        if tface:
            assert cfg.model.max_face_length == tface
        if tseq:
            assert cfg.model.label_seq_length == tseq
    assert load_cfg(os.path.join(ROOT, "configs", "seq2seq.yml")).post_process.is_coedge is False


def test_config_rejects_unknown_keys_and_bad_types():
    from faceformer_amd.config import load_cfg
    with pytest.raises(KeyError):
        load_cfg("", ["model.not_a_key", "1"])
    with pytest.raises(ValueError):
        load_cfg("", ["model.num_lines", "abc"])
    with pytest.raises(ValueError):
        load_cfg("", ["model.num_lines"])
    assert load_cfg("", ["trainer.lr", "1"]).trainer.lr == 1.0  # int -> float promotion


# ---- module surface / state_dict --------------------------------------------------
def test_state_dict_layout_matches_appendix_b():
    from faceformer_amd.config import load_cfg
    from faceformer_amd.models import SurfaceFormer, SurfaceFormer_Parallel
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"))
    m = SurfaceFormer_Parallel(**cfg.model)
    spec = state_dict_spec("parallel", 216, 37)
    sd = m.state_dict()
    assert len(sd) == 195 and list(sd.keys()) == [s[0] for s in spec]
    for name, shape, _ in spec:
        assert tuple(sd[name].shape) == tuple(shape), name
    assert sum(p.numel() for p in m.parameters()) == 32256000
    m.load_state_dict(make_state_dict(spec, "gain4", 1))       # strict load of synthetic weights
    cfg2 = load_cfg(os.path.join(ROOT, "configs", "seq2seq+coedge.yml"))
    m2 = SurfaceFormer(**cfg2.model)
    assert sum(p.numel() for p in m2.parameters()) == 32369664
    assert list(m2.state_dict().keys()) == [s[0] for s in state_dict_spec("seq2seq", 216, 259)]
    # a Lightning checkpoint prefixes every key with "model." (reference trainer.py:20)
    pref = {"model." + k: v for k, v in m2.state_dict().items()}
    m2.load_state_dict({k[len("model."):]: v for k, v in pref.items()})


def test_import_surface_of_reference_modules():
    import faceformer_amd.embedding as emb
    import faceformer_amd.transformer as tr
    import faceformer_amd.utils as ut
    for name in ("Transformer", "TransformerEncoder", "TransformerDecoder", "TransformerEncoderLayer",
                 "TransformerDecoderLayer", "_get_clones", "_get_activation_fn"):
        assert hasattr(tr, name)
    for name in ("VanillaEmedding", "CoordinateEmbedding", "PositionalEncoding", "PositionEmbeddingLearned"):
        assert hasattr(emb, name)
    assert ut.min_value_of_dtype(torch.float32) == torch.finfo(torch.float32).min
    assert ut.max_value_of_dtype(torch.int32) == 2 ** 31 - 1
    assert ut.tiny_value_of_dtype(torch.half) == 1e-4 and ut.flatten_list([[1], [2, 3]]) == [1, 2, 3]
    with pytest.raises(TypeError):
        ut.min_value_of_dtype(torch.bool)
    layer = tr.TransformerDecoderLayer(128, 2, 256, 0.1, "relu", True)
    assert {n for n, _ in layer.named_parameters()} >= {"self_attn.in_proj_weight", "multihead_attn.out_proj.bias",
                                                        "linear1.weight", "norm3.bias"}
    pe = emb.PositionEmbeddingLearned(64, max_len=10)
    assert tuple(pe(torch.zeros(2, 7, 64)).shape) == (1, 7, 64)
    with pytest.raises(RuntimeError):
        tr._get_activation_fn("swish")


def test_attn_mask_forms_are_validated_on_the_host():
    """torch's `attn_mask` forms (2-D [L, S] / 3-D [N*H, L, S], boolean or floating point) are sorted into (causal, additive bias,
    boolean mask) before any HIP call; the 'subsequent' triangle (reference model.py:71-73) is recognised as the causal rule of the
    MFMA kernels; wrong shapes / ranks / dtypes are ValueError / TypeError like torch's own checks."""
    from faceformer_amd import transformer as tr
    L, S, BH = 5, 7, 6
    assert tr._attn_mask_forms(None, "m", L, S, BH) == (False, None, None)
    tri = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)
    assert tr._attn_mask_forms(tri, "m", L, L, BH) == (True, None, None)
    assert tr._attn_mask_forms(tri.to(torch.uint8), "m", L, L, BH)[0] is True
    causal, bias, mask = tr._attn_mask_forms(~tri, "m", L, L, BH)
    assert causal is False and bias is None and mask.dtype == torch.uint8 and tuple(mask.shape) == (L, L)
    causal, bias, mask = tr._attn_mask_forms(torch.zeros(BH, L, S, dtype=torch.float64), "m", L, S, BH)
    assert causal is False and mask is None and bias.dtype == torch.float32 and tuple(bias.shape) == (BH, L, S)
    with pytest.raises(ValueError):
        tr._attn_mask_forms(torch.zeros(L, S + 1), "m", L, S, BH)
    with pytest.raises(ValueError):
        tr._attn_mask_forms(torch.zeros(BH + 1, L, S), "m", L, S, BH)
    with pytest.raises(ValueError):
        tr._attn_mask_forms(torch.zeros(1, BH, L, S), "m", L, S, BH)
    with pytest.raises(TypeError):
        tr._attn_mask_forms(torch.zeros(L, S, dtype=torch.int32), "m", L, S, BH)
    with pytest.raises(TypeError):
        tr._mask_u8(torch.zeros(2, S, dtype=torch.int64), "key_padding_mask", (2, S))
    assert tr._mask_u8(torch.zeros(2, S), "key_padding_mask", (2, S)).dtype == torch.float32     # additive padding mask
    mha = tr.MultiheadAttention(96, 2)                                                           # 48-wide heads: constructible, CPU run refused
    assert mha.head_dim == 48
    from faceformer_amd.hip.lib import HipExtensionError
    with pytest.raises(HipExtensionError):
        mha.eval()(torch.zeros(3, 1, 96), torch.zeros(4, 1, 96), torch.zeros(4, 1, 96))


def test_alias_package_exposes_reference_names():
    import faceformer
    from faceformer.models import SurfaceFormer, SurfaceFormer_Parallel  # noqa: F401
    from faceformer.transformer import TransformerDecoderLayer  # noqa: F401
    from faceformer.embedding import VanillaEmedding  # noqa: F401
    from faceformer.config import get_cfg, get_parser  # noqa: F401
    from faceformer.utils import min_value_of_dtype  # noqa: F401
    import faceformer_amd.models
    assert SurfaceFormer is faceformer_amd.models.SurfaceFormer
    assert faceformer.__name__ == "faceformer"


# ---- no CPU fallback ------------------------------------------------------------------------------------
def test_cpu_forward_fails_loudly():
    from faceformer_amd.hip.lib import HipExtensionError
    from faceformer_amd.models import SurfaceFormer_Parallel
    from faceformer_amd import transformer as tr
    m = SurfaceFormer_Parallel(num_model=128, num_head=2, num_feedforward=256, num_encoder_layers=1,
                               num_decoder_layers=1, num_lines=8, max_face_length=5, token=token_ns()).eval()
    batch = dict(input=torch.zeros(1, 8, 50, 2), input_mask=torch.zeros(1, 8, dtype=torch.bool),
                 label=torch.zeros(1, 8, 5, dtype=torch.long), num_input=[8])
    with pytest.raises(HipExtensionError):
        m(batch)
    with pytest.raises(NotImplementedError):
        m.train()(batch)
    enc = tr.TransformerEncoderLayer(128, 2, 256, 0.0, "relu", True).eval()
    with pytest.raises(HipExtensionError):
        enc(torch.zeros(4, 1, 128))


def test_product_code_never_imports_the_oracle():
    bad = []
    for path in glob.glob(os.path.join(ROOT, "faceformer_amd", "**", "*.py"), recursive=True) + \
            glob.glob(os.path.join(ROOT, "faceformer", "**", "*.py"), recursive=True):
        src = open(path).read()
        if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) or "/root/reference" in src:
            bad.append(path)
    assert not bad, bad


# ---- C ABI ------------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(hip_lib):
    header = open(os.path.join(ROOT, "include", "faceformer_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(ff_[a-z0-9_]+)\s*\(", header))
    assert {"ff_gemm_f32", "ff_attention", "ff_pointer_argmax", "ff_layernorm", "ff_encode", "ff_decode",
            "ff_version"} <= declared
    from faceformer_amd.hip import lib
    assert declared == set(lib.SIGNATURES), (declared ^ set(lib.SIGNATURES))
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert hip_lib.ff_version() >= 100
    assert hip_lib.ff_last_error() is not None
    assert hip_lib.ff_device_count() >= 0


def test_struct_layouts_match_the_header(hip_lib):
    """ctypes mirrors vs. the compiler's layout, via a tiny C probe compiled with gcc."""
    import shutil
    import subprocess
    import tempfile
    from faceformer_amd.hip import lib
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    probe = r'''
#include <stdio.h>
#include "faceformer_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ff_attn_desc), sizeof(ff_mha_weights), sizeof(ff_layer_weights),
         sizeof(ff_model), sizeof(ff_decode_params), offsetof(ff_model, dec), sizeof(ff_gemm_ln_desc),
         offsetof(ff_gemm_ln_desc, ln_stats_out), sizeof(ff_attn_general_desc), offsetof(ff_attn_general_desc, attn_batch_stride),
         offsetof(ff_attn_general_desc, scale));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "p.c"), os.path.join(d, "p")
        open(src, "w").write(probe)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    got = [ctypes.sizeof(lib.AttnDesc), ctypes.sizeof(lib.MhaWeights), ctypes.sizeof(lib.LayerWeights),
           ctypes.sizeof(lib.Model), ctypes.sizeof(lib.DecodeParams), lib.Model.dec.offset,
           ctypes.sizeof(lib.GemmLnDesc), lib.GemmLnDesc.ln_stats_out.offset, ctypes.sizeof(lib.AttnGeneralDesc),
           lib.AttnGeneralDesc.attn_batch_stride.offset, lib.AttnGeneralDesc.scale.offset]
    assert [int(x) for x in out] == got


def test_entry_points_reject_bad_arguments_without_touching_the_device(hip_lib):
    """Argument validation happens before any HIP call: it works (and is tested) on a host without a GPU;
    the error text names the offending argument."""
    from faceformer_amd.hip import lib
    FF_ERR_ARG = -1
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p += (-p) % 16 or 0
    # K not a multiple of 4 / null operand / bad tile id
    assert hip_lib.ff_gemm_f32(p, 8, None, 0, p, 8, None, None, 0, p, 8, 2, 2, 6, 0, 0, None) == FF_ERR_ARG
    assert b"K" in hip_lib.ff_last_error()
    assert hip_lib.ff_gemm_f32(None, 8, None, 0, p, 8, None, None, 0, p, 8, 2, 2, 8, 0, 0, None) == FF_ERR_ARG
    assert hip_lib.ff_gemm_f32(p, 8, None, 0, p, 8, None, None, 0, p, 8, 2, 2, 8, 0, 99, None) == FF_ERR_ARG
    # empty problems are a no-op, not an error
    assert hip_lib.ff_gemm_f32(p, 8, None, 0, p, 8, None, None, 0, p, 8, 0, 2, 8, 0, 0, None) == 0
    # LayerNorm width must be a multiple of 4; attention needs its tensors
    assert hip_lib.ff_layernorm(p, 8, p, p, 1e-5, p, 8, None, 0, None, 0, 1, 1, 2, 6, None) == FF_ERR_ARG
    d = lib.AttnDesc()
    d.num_groups, d.num_heads, d.nq, d.nk = 1, 1, 4, 4
    assert hip_lib.ff_attention(ctypes.byref(d), None) == FF_ERR_ARG
    assert hip_lib.ff_attention(None, None) == FF_ERR_ARG
    ga = lib.AttnGeneralDesc()
    ga.num_groups, ga.num_heads, ga.head_dim, ga.nq, ga.nk = 1, 1, 0, 4, 4
    assert hip_lib.ff_attention_general(ctypes.byref(ga), None) == FF_ERR_ARG and b"head_dim" in hip_lib.ff_last_error()
    ga.head_dim = 32
    assert hip_lib.ff_attention_general(ctypes.byref(ga), None) == FF_ERR_ARG and b"null tensor" in hip_lib.ff_last_error()
    assert hip_lib.ff_attention_general(None, None) == FF_ERR_ARG
    # the bf16-split product: K must be a multiple of 32 and at least 64
    assert hip_lib.ff_gemm_x3(p, 48, None, 0, p, None, None, 0, p, 8, 2, 2, 48, 0, None) == FF_ERR_ARG
    assert b"K" in hip_lib.ff_last_error()
    # the LayerNorm-fused form: statistics must describe whole rows, a row table needs statistics
    g = lib.GemmLnDesc()
    g.A, g.lda, g.W, g.ldw, g.C, g.ldc, g.M, g.N, g.K = p, 512, p, 512, p, 512, 64, 512, 512
    g.ln_stats_in, g.ln_nseg = p, 8
    assert hip_lib.ff_gemm_f32_ln(ctypes.byref(g), None) == FF_ERR_ARG
    g.ln_stats_in, g.ln_nseg, g.row_table, g.row_div, g.row_cols, g.ld_row_table = None, 0, p, 4, 512, 512
    assert hip_lib.ff_gemm_f32_ln(ctypes.byref(g), None) == FF_ERR_ARG
    assert hip_lib.ff_set_gemm_tuning(0, 1, 1, 1) == FF_ERR_ARG


def test_tuning_table_is_one_settable_struct(hip_lib):
    """Round 6: every A/B knob of the library is one int in ONE table, initialised from the environment
    variable of the same name and settable per process (ff_set_tuning / ff_get_tuning / ff_reset_tuning) -- no function-local
    statics frozen on first use.  No GPU needed: the table is host state."""
    FF_ERR_ARG = -1
    v = ctypes.c_int(-7)
    defaults = {b"FF_L0_FOLD": 1, b"FF_POINTER_FOLD": 1, b"FF_LAST_QKV_ONE_LAUNCH_ROWS": 512, b"FF_PINNED_COUNTERS": 65536,
                b"FF_DMA_MIN_ROWS": 4096, b"FF_DMA_MIN_ROWS_N512": 7680, b"FF_DMA_MIN_ROWS_WIDE": 2560, b"FF_SK_HYBRID": 1,
                b"FF_SK_HYBRID_FIX": 10, b"FF_SK_HYBRID_MAXLEFT8": 4, b"FF_SK_HYBRID_MINU": 2, b"FF_SK_HYBRID_FORCE": 0,
                b"FF_NO_PANEL": 0, b"FF_X3_SMALL_SPLIT": 0, b"FF_RK_SPLIT_OLD": 1, b"FF_RK_SPLIT_YOUNG": 1, b"FF_RK_PHASE": 0,
                b"FF_RK_ROTATE": 1, b"FF_DEBUG_TIMING": 0, b"FF_X3_NEED_N1024": 7, b"FF_X3_NEED_N512": 11, b"FF_X2H_ATTN": 1}
    assert hip_lib.ff_reset_tuning() == 0
    for name, want in defaults.items():
        assert hip_lib.ff_get_tuning(name, ctypes.byref(v)) == 0 and v.value == want, name
    try:
        assert hip_lib.ff_set_tuning(b"FF_DMA_MIN_ROWS", 123) == 0
        assert hip_lib.ff_get_tuning(b"FF_DMA_MIN_ROWS", ctypes.byref(v)) == 0 and v.value == 123
        assert hip_lib.ff_set_tuning(b"FF_NOT_A_KNOB", 1) == FF_ERR_ARG and b"FF_NOT_A_KNOB" in hip_lib.ff_last_error()
        assert hip_lib.ff_get_tuning(b"FF_NOT_A_KNOB", ctypes.byref(v)) == FF_ERR_ARG
        assert hip_lib.ff_set_tuning(b"FF_PINNED_COUNTERS", 0) == FF_ERR_ARG          # must stay inside the allocated slots
        assert hip_lib.ff_set_tuning(None, 1) == FF_ERR_ARG
    finally:
        assert hip_lib.ff_reset_tuning() == 0
    assert hip_lib.ff_get_tuning(b"FF_DMA_MIN_ROWS", ctypes.byref(v)) == 0 and v.value == 4096
    # the sources read no environment variable anywhere else
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hits = [f for f in glob.glob(os.path.join(root, "faceformer_amd", "csrc", "*")) if "getenv" in open(f).read()]
    assert [os.path.basename(f) for f in hits] == ["ff_rowops.hip"], hits
