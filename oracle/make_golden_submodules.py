"""ORACLE TOOLING (build container only): golden vectors for the SUB-MODULE surface of SURVEY.md 8(b) --
`faceformer.transformer.{TransformerEncoder, TransformerDecoder, *Layer, Transformer}` called the way an external
caller (the reference's own `forward_train`, model_para.py:99-171, or any user of the blocks) calls them, and
`faceformer.embedding.*` -- captured from the IMPORTED reference classes, and the pin of the matching
restatements in `oracle/refpath.py` (bit for bit).

Run:  python oracle/make_golden_submodules.py      -> tests/golden/submodule_cases.npz

Weights and inputs are regenerated from names + seeds (`faceformer_amd.synth.make_module_state`,
`make_named_tensor`): the fixture holds outputs only.  /root/reference does not exist on the GPU box; nothing at
test / bench time imports this script.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"

E, H, FF = 128, 2, 256

# name -> description of one call; shared with tests/test_submodule_surface.py through the fixture's JSON header
CASES = {
    # teacher-forced caller: causal tgt_mask + tgt_key_padding_mask + memory mask (model_para.py:125,162-163)
    "dec_pre_causal": dict(kind="decoder", pre=True, layers=2, t=7, B=5, S=11, causal=True, tgt_kpm=True,
                           mem_kpm=True, intermediate=False, seed=1),
    # the eval call: no tgt masks (model_para.py:222-223)
    "dec_pre_plain": dict(kind="decoder", pre=True, layers=2, t=6, B=4, S=9, causal=False, tgt_kpm=False,
                          mem_kpm=True, intermediate=False, seed=2),
    "dec_pre_intermediate": dict(kind="decoder", pre=True, layers=2, t=3, B=2, S=8, causal=False, tgt_kpm=False,
                                 mem_kpm=False, intermediate=True, seed=3),
    "dec_post": dict(kind="decoder", pre=False, layers=2, t=5, B=3, S=10, causal=False, tgt_kpm=False,
                     mem_kpm=True, intermediate=False, seed=4),
    "dec_post_causal": dict(kind="decoder", pre=False, layers=1, t=6, B=3, S=7, causal=True, tgt_kpm=True,
                            mem_kpm=True, intermediate=False, seed=5),
    "enc_pre": dict(kind="encoder", pre=True, layers=2, S=13, B=3, kpm=True, seed=6),
    "enc_post": dict(kind="encoder", pre=False, layers=2, S=12, B=4, kpm=True, seed=7),
    # activation="gelu" (transformer.py:276-284): a constructor keyword of the boundary that no reference config sets
    "enc_pre_gelu": dict(kind="encoder", pre=True, layers=2, S=10, B=3, kpm=True, seed=14, act="gelu"),
    "dec_post_gelu": dict(kind="decoder", pre=False, layers=1, t=4, B=3, S=9, causal=False, tgt_kpm=False,
                          mem_kpm=True, intermediate=False, seed=15, act="gelu"),
    # round 6 -- head widths other than 64 (transformer.py:131,191-192 hand num_model / num_head to nn.MultiheadAttention as they
    # come) and torch's general attn_mask forms (transformer.py:70-73 `mask` -> src_mask; :95-101 tgt_mask / memory_mask): boolean
    # or additive float, one [L, S] matrix or one per (batch, head).  All of these run on ff_attention_general.
    "enc_pre_h32": dict(kind="encoder", pre=True, layers=2, S=13, B=3, kpm=True, seed=16, H=4),
    "dec_pre_h128_causal": dict(kind="decoder", pre=True, layers=2, t=7, B=4, S=11, causal=True, tgt_kpm=True, mem_kpm=True,
                                intermediate=False, seed=17, H=1),
    "dec_post_h16": dict(kind="decoder", pre=False, layers=1, t=5, B=3, S=70, causal=False, tgt_kpm=False, mem_kpm=True,
                         intermediate=False, seed=18, H=8),
    "dec_pre_h48": dict(kind="decoder", pre=True, layers=1, t=4, B=2, S=9, causal=False, tgt_kpm=False, mem_kpm=True,
                        intermediate=False, seed=19, H=2, E=96, FF=192),
    "enc_pre_srcmask_float": dict(kind="encoder", pre=True, layers=2, S=12, B=3, kpm=True, seed=20, src_mask="float"),
    "enc_post_srcmask_bool3d": dict(kind="encoder", pre=False, layers=1, S=9, B=2, kpm=False, seed=21, src_mask="bool3d"),
    "dec_pre_tgtmask_band": dict(kind="decoder", pre=True, layers=2, t=8, B=3, S=10, causal=False, tgt_kpm=False, mem_kpm=True,
                                 intermediate=False, seed=22, tgt_mask="band"),
    "dec_pre_memmask_float": dict(kind="decoder", pre=True, layers=2, t=6, B=3, S=12, causal=True, tgt_kpm=False, mem_kpm=True,
                                  intermediate=False, seed=23, memory_mask="float"),
    "dec_post_memmask_bool_tgt_float3d": dict(kind="decoder", pre=False, layers=1, t=5, B=2, S=8, causal=False, tgt_kpm=True,
                                              mem_kpm=False, intermediate=False, seed=24, memory_mask="bool", tgt_mask="float3d"),
    # DETR-style wrapper that neither model class instantiates (transformer.py:18-59)
    "transformer_post": dict(kind="transformer", pre=False, enc=1, dec=2, N=2, Hh=3, Ww=4, Q=5, intermediate=True,
                             seed=8),
    "transformer_pre": dict(kind="transformer", pre=True, enc=1, dec=1, N=2, Hh=2, Ww=5, Q=4, intermediate=False,
                            seed=9),
    "vanilla_embedding": dict(kind="vanilla", N=3, L=6, seed=10),
    "position_tables": dict(kind="positions", seed=11),
    "coordinate_embedding": dict(kind="coordinate", N=2, S=5, seed=12),
    # the pointer head with the reference's argument layout: embedding S x B x E, pointer t x B x E, mask B x S
    "select_next": dict(kind="select_next", S=19, B=6, t=3, seed=13),
}


def _import_reference():
    pkg = types.ModuleType("faceformer")
    pkg.__path__ = [os.path.join(REFERENCE, "faceformer")]
    sys.modules["faceformer"] = pkg
    import faceformer.embedding as ref_emb
    import faceformer.models as ref_models
    import faceformer.transformer as ref_tr
    return ref_tr, ref_emb, ref_models


def _general_mask(name, kind, lq, lk, batch_heads, seed):
    """torch attn_mask forms: "float" additive [lq, lk]; "float3d" additive [batch*heads, lq, lk]; "bool" / "bool3d" removal masks
    that keep key 0 of every query (a query without keys is NaN in torch: not a comparison case); "band": |i - j| > 2 removed."""
    from faceformer_amd.synth import make_named_tensor as T
    if kind == "band":
        i = torch.arange(lq)[:, None]
        j = torch.arange(lk)[None, :]
        return (i - j).abs() > 2
    shape = (batch_heads, lq, lk) if kind.endswith("3d") else (lq, lk)
    x = T(name, shape, seed, 2.0)
    if kind.startswith("float"):
        return x
    m = x > 0.6
    m[..., 0] = False
    return m


def make_inputs(name, c):
    """Seeded inputs of a case (the test regenerates them with this function's twin in the test file)."""
    from faceformer_amd.synth import make_named_tensor as T
    s = c["seed"]
    E, H = c.get("E", globals()["E"]), c.get("H", globals()["H"])
    if c["kind"] == "decoder":
        d = dict(tgt=T(name + ".tgt", (c["t"], c["B"], E), s), memory=T(name + ".memory", (c["S"], c["B"], E), s),
                 pos=T(name + ".pos", (c["S"], 1, E), s, 0.5), query_pos=T(name + ".qpos", (c["t"], 1, E), s, 0.5))
        if c["causal"]:
            d["tgt_mask"] = torch.triu(torch.ones(c["t"], c["t"], dtype=torch.bool), diagonal=1)
        if c.get("tgt_mask"):
            d["tgt_mask"] = _general_mask(name + ".tgt_mask", c["tgt_mask"], c["t"], c["t"], c["B"] * H, s)
        if c.get("memory_mask"):
            d["memory_mask"] = _general_mask(name + ".memory_mask", c["memory_mask"], c["t"], c["S"], c["B"] * H, s)
        if c["tgt_kpm"]:   # trailing positions padded, never position 0 (a causal row must keep one key)
            keep = 2 + (torch.arange(c["B"]) * 3) % (c["t"] - 1)
            d["tgt_key_padding_mask"] = torch.arange(c["t"])[None, :] >= keep[:, None]
        if c["mem_kpm"]:
            keep = 3 + (torch.arange(c["B"]) * 5) % (c["S"] - 2)
            d["memory_key_padding_mask"] = torch.arange(c["S"])[None, :] >= keep[:, None]
        return d
    if c["kind"] == "encoder":
        d = dict(src=T(name + ".src", (c["S"], c["B"], E), s), pos=T(name + ".pos", (c["S"], 1, E), s, 0.5))
        if c["kpm"]:
            keep = 4 + (torch.arange(c["B"]) * 4) % (c["S"] - 3)
            d["src_key_padding_mask"] = torch.arange(c["S"])[None, :] >= keep[:, None]
        if c.get("src_mask"):
            d["src_mask"] = _general_mask(name + ".src_mask", c["src_mask"], c["S"], c["S"], c["B"] * H, s)
        return d
    if c["kind"] == "transformer":
        hw = c["Hh"] * c["Ww"]
        keep = hw - (torch.arange(c["N"]) * 3) % 5
        return dict(src=T(name + ".src", (c["N"], E, c["Hh"], c["Ww"]), s),
                    mask=(torch.arange(hw)[None, :] >= keep[:, None]).view(c["N"], c["Hh"], c["Ww"]),
                    query_embed=T(name + ".query", (c["Q"], E), s, 0.5),
                    pos_embed=T(name + ".pos", (c["N"], E, c["Hh"], c["Ww"]), s, 0.5))
    if c["kind"] == "vanilla":
        return dict(coord=T(name + ".coord", (c["N"], c["L"], 50, 2), s))
    if c["kind"] == "coordinate":
        g = np.random.default_rng([77, s])
        return dict(coord=torch.from_numpy(g.integers(0, 2 ** 4, size=(c["N"], c["S"], 2))))
    if c["kind"] == "select_next":
        keep = 6 + (torch.arange(c["B"]) * 7) % (c["S"] - 5)
        return dict(embedding=T(name + ".embedding", (c["S"], c["B"], E), s),
                    pointer=T(name + ".pointer", (c["t"], c["B"], E), s),
                    input_mask=torch.arange(c["S"])[None, :] >= keep[:, None])
    return {}


def build(name, c, tr, emb, models, token):
    """The module under test, constructed through the PUBLIC constructors (works for the reference's modules and
    for faceformer_amd's: same signatures)."""
    E, H, FF = c.get("E", globals()["E"]), c.get("H", globals()["H"]), c.get("FF", globals()["FF"])
    if c["kind"] == "decoder":
        layer = tr.TransformerDecoderLayer(E, H, FF, 0.1, c.get("act", "relu"), c["pre"])
        return tr.TransformerDecoder(layer, c["layers"], torch.nn.LayerNorm(E), return_intermediate=c["intermediate"])
    if c["kind"] == "encoder":
        layer = tr.TransformerEncoderLayer(E, H, FF, 0.1, c.get("act", "relu"), c["pre"])
        return tr.TransformerEncoder(layer, c["layers"], torch.nn.LayerNorm(E) if c["pre"] else None)
    if c["kind"] == "transformer":
        return tr.Transformer(num_model=E, num_head=H, num_encoder_layers=c["enc"], num_decoder_layers=c["dec"],
                              num_feedforward=FF, dropout=0.1, activation="relu", normalize_before=c["pre"],
                              return_intermediate_dec=c["intermediate"])
    if c["kind"] == "vanilla":
        return emb.VanillaEmedding(100, E, token)
    if c["kind"] == "coordinate":
        return emb.CoordinateEmbedding(2, 4, 16, E)
    if c["kind"] == "select_next":
        return models.SurfaceFormer_Parallel(num_model=E, num_head=H, num_feedforward=FF, num_encoder_layers=1,
                                             num_decoder_layers=1, num_lines=c["S"] - 4, max_face_length=5, token=token)
    raise KeyError(c["kind"])


def load_weights(module, name, c):
    from faceformer_amd.synth import make_module_state
    sd = make_module_state({k: v.shape for k, v in module.state_dict().items()}, seed=c["seed"])
    module.load_state_dict(sd)
    return sd


def call(module, c, inp):
    if c["kind"] == "decoder":
        kw = {k: inp[k] for k in ("tgt_mask", "memory_mask", "tgt_key_padding_mask", "memory_key_padding_mask") if k in inp}
        return module(inp["tgt"], inp["memory"], pos=inp["pos"], query_pos=inp["query_pos"], **kw)
    if c["kind"] == "encoder":
        return module(inp["src"], mask=inp.get("src_mask"), src_key_padding_mask=inp.get("src_key_padding_mask"), pos=inp["pos"])
    if c["kind"] == "transformer":
        hs, mem = module(inp["src"], inp["mask"], inp["query_embed"], inp["pos_embed"])
        return hs, mem
    if c["kind"] in ("vanilla", "coordinate"):
        return module(inp["coord"])
    if c["kind"] == "select_next":
        return module.select_next(inp["embedding"], inp["pointer"], inp["input_mask"])
    raise KeyError(c["kind"])


def restate(name, c, sd, inp):
    """The same call through oracle/refpath.py's functional restatements (pinned here, bit for bit)."""
    from oracle import refpath as R
    H = c.get("H", globals()["H"])
    if c["kind"] == "decoder":
        return R.decoder_stack(sd, "", inp["tgt"], inp["memory"], H, c["layers"], normalize_before=c["pre"],
                               final_norm=True, return_intermediate=c["intermediate"], tgt_mask=inp.get("tgt_mask"),
                               tgt_key_padding_mask=inp.get("tgt_key_padding_mask"),
                               memory_key_padding_mask=inp.get("memory_key_padding_mask"), pos=inp["pos"],
                               query_pos=inp["query_pos"], activation=c.get("act", "relu"), memory_mask=inp.get("memory_mask"))
    if c["kind"] == "encoder":
        return R.encoder_stack(sd, "", inp["src"], H, c["layers"], normalize_before=c["pre"], final_norm=c["pre"],
                               src_key_padding_mask=inp.get("src_key_padding_mask"), pos=inp["pos"], activation=c.get("act", "relu"),
                               src_mask=inp.get("src_mask"))
    if c["kind"] == "select_next":
        return R.select_next(inp["embedding"], inp["pointer"], inp["input_mask"])[0]
    return None


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("make_golden_submodules.py needs /root/reference (build container only)")
    sys.path.insert(0, ROOT)
    tr, emb, models = _import_reference()
    token = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    payload = {"cases": np.frombuffer(json.dumps(dict(E=E, H=H, FF=FF, cases=CASES), sort_keys=True).encode(), dtype=np.uint8)}
    torch.manual_seed(0)
    for name, c in CASES.items():
        if c["kind"] == "positions":
            pe = emb.PositionEmbeddingLearned(E, max_len=9)
            from faceformer_amd.synth import make_module_state
            pe.load_state_dict(make_module_state({k: v.shape for k, v in pe.state_dict().items()}, seed=c["seed"]))
            payload[name + "/learned"] = pe(torch.zeros(3, 6, E)).detach().numpy()
            payload[name + "/sinusoid"] = emb.PositionalEncoding(E, max_len=20)(torch.zeros(2, 7, E)).numpy()
            print("  %-22s learned %s sinusoid %s" % (name, payload[name + "/learned"].shape, payload[name + "/sinusoid"].shape))
            continue
        module = build(name, c, tr, emb, models, token).eval()
        sd = load_weights(module, name, c)
        inp = make_inputs(name, c)
        with torch.no_grad():
            out = call(module, c, inp)
            rs = restate(name, c, sd, inp)
        outs = out if isinstance(out, tuple) else (out,)
        if rs is not None:
            assert torch.equal(rs, outs[0]), "%s: oracle restatement differs from the imported reference" % name
        for i, o in enumerate(outs):
            payload["%s/out%d" % (name, i)] = o.numpy()
        if c["kind"] == "select_next":   # margins of the reference's logits: the test requires equal tokens at decisive ones
            logit = torch.bmm(inp["embedding"].transpose(0, 1), inp["pointer"].permute(1, 2, 0)[..., -1:]).squeeze(-1)
            logit = logit.masked_fill(inp["input_mask"], torch.finfo(torch.float32).min)
            v = torch.sort(logit, dim=1, descending=True).values
            payload[name + "/margin"] = (v[:, 0] - v[:, 1]).numpy()
            payload[name + "/logits"] = logit.numpy()
        print("  %-22s %s  max|out| %.3g  %s" % (name, [tuple(o.shape) for o in outs], float(outs[0].abs().max()) if outs[0].is_floating_point() else -1,
                                              "restatement bit-identical" if rs is not None else ""))
    path = os.path.join(ROOT, "tests", "golden", "submodule_cases.npz")
    np.savez_compressed(path, **payload)
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
