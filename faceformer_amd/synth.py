"""Seeded synthetic wireframes and build-owned, name-keyed synthetic weights.

No dataset samples or trained checkpoints are reachable (reference README.md:33,38 point at Google
Drive), so parity and throughput are both established on synthetic inputs that either side (the HIP
path, the oracle, the golden-vector generator) can regenerate from `(recipe, seed)` alone:

* `state_dict_spec(...)`  -- names/shapes of the 195 tensors of SURVEY.md Appendix B;
* `make_state_dict(...)`  -- numpy-PRNG weights keyed by parameter *name* (crc32(name) ^ seed), so
  tensors do not depend on construction order or on torch's generator;
* `make_wireframes(...)`  -- straight-segment wireframes sampled to 50 points exactly like the
  reference data loader does for two-point edges (reference `datasets/data_para.py:14-19`), sorted
  lexicographically, zero-padded to `num_lines`, with the reference's batch keys
  (`input`, `input_mask`, `label`, `num_input`; reference `datasets/data_para.py:99-110`).
"""
import zlib

import numpy as np
import torch

RECIPES = ("default", "gain4", "bias05")


def state_dict_spec(kind, num_lines, seq_len, num_model=512, num_feedforward=1024,
                    num_encoder_layers=6, num_decoder_layers=6, in_dim=100, num_token=4, encoder_norm=True):
    """[(name, shape, role)] in the reference's registration order. role in
    {'matrix','ln_w','ln_b','lin_b','attn_b','buffer'}; `kind` is 'parallel' or 'seq2seq' (same
    tensors, `seq_len` = max_face_length or label_seq_length)."""
    E, FF = num_model, num_feedforward
    S = num_lines + num_token
    spec = [
        ("val_enc.embedding_token.weight", (num_token, E), "matrix"),
        ("val_enc.embedding_value.0.weight", (E, in_dim), "matrix"),
        ("val_enc.embedding_value.0.bias", (E,), "lin_b:%d" % in_dim),
        ("val_enc.embedding_value.2.weight", (E, E), "matrix"),
        ("val_enc.embedding_value.2.bias", (E,), "lin_b:%d" % E),
        ("pos_enc.position", (1, S), "buffer"),
        ("pos_enc.pos_embed.weight", (S, E), "matrix"),
        ("query_pos_enc.position", (1, seq_len), "buffer"),
        ("query_pos_enc.pos_embed.weight", (seq_len, E), "matrix"),
    ]

    def attn(p):
        return [(p + ".in_proj_weight", (3 * E, E), "matrix"), (p + ".in_proj_bias", (3 * E,), "attn_b"),
                (p + ".out_proj.weight", (E, E), "matrix"), (p + ".out_proj.bias", (E,), "attn_b")]

    def ffn(p):
        return [(p + ".linear1.weight", (FF, E), "matrix"), (p + ".linear1.bias", (FF,), "lin_b:%d" % E),
                (p + ".linear2.weight", (E, FF), "matrix"), (p + ".linear2.bias", (E,), "lin_b:%d" % FF)]

    def norm(p):
        return [(p + ".weight", (E,), "ln_w"), (p + ".bias", (E,), "ln_b")]

    for i in range(num_encoder_layers):
        p = "encoder.layers.%d" % i
        spec += attn(p + ".self_attn") + ffn(p) + norm(p + ".norm1") + norm(p + ".norm2")
    if encoder_norm:      # (a post-norm model has no final encoder norm: reference model.py:36)
        spec += norm("encoder.norm")
    for i in range(num_decoder_layers):
        p = "decoder.layers.%d" % i
        spec += (attn(p + ".self_attn") + attn(p + ".multihead_attn") + ffn(p)
                 + norm(p + ".norm1") + norm(p + ".norm2") + norm(p + ".norm3"))
    spec += norm("decoder.norm")
    spec += [("project.weight", (E, E), "matrix"), ("project.bias", (E,), "lin_b:%d" % E)]
    return spec


def _rng(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()) & 0xFFFFFFFF, int(seed) & 0xFFFFFFFF])


def make_state_dict(spec, recipe="default", seed=0):
    """Name-keyed synthetic weights (float32 torch tensors on CPU).

    'default': what the reference's constructors leave behind in distribution -- every tensor with
        dim > 1 xavier-uniform (reference model.py:49-52), Linear biases U(+-1/sqrt(fan_in)), attention
        biases 0, LayerNorm 1/0.  Greedy decode never stops early with it (throughput weights).
    'gain4'  : matrices x4, LayerNorm weights 1+0.3 N(0,1), all other vectors 0.3 N(0,1): spreads the
        pointer distribution so that different anchors decode different loops (parity weights).
    'bias05' : default matrices, every vector 0.5 N(0,1) (LayerNorm weights 1+0.5 N(0,1)): drives all
        sequences to a special token at once (exercises the early-break + zero-pad branch).
    """
    if recipe not in RECIPES:
        raise ValueError("unknown weight recipe %r" % (recipe,))
    sd = {}
    for name, shape, role in spec:
        g = _rng(name, seed)
        if role == "buffer":
            sd[name] = torch.arange(shape[1], dtype=torch.long).unsqueeze(0)
            continue
        if role == "matrix":
            fan_out, fan_in = shape[0], shape[1]
            bound = np.sqrt(6.0 / (fan_in + fan_out))
            w = g.uniform(-bound, bound, size=shape)
            if recipe == "gain4":
                w = w * 4.0
        elif role == "ln_w":
            w = np.ones(shape)
            if recipe == "gain4":
                w = 1.0 + 0.3 * g.standard_normal(shape)
            elif recipe == "bias05":
                w = 1.0 + 0.5 * g.standard_normal(shape)
        else:  # ln_b, lin_b:<fan_in>, attn_b
            if recipe == "gain4":
                w = 0.3 * g.standard_normal(shape)
            elif recipe == "bias05":
                w = 0.5 * g.standard_normal(shape)
            elif role.startswith("lin_b"):
                b = 1.0 / np.sqrt(float(role.split(":")[1]))
                w = g.uniform(-b, b, size=shape)
            else:
                w = np.zeros(shape)
        sd[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return sd


def _segment_points(p0, p1, num_points):
    t = np.linspace(0, 1, num_points)
    x = p0[0] + (p1[0] - p0[0]) * t
    y = p0[1] + (p1[1] - p0[1]) * t
    return np.vstack([x, y]).T


def make_wireframes(num_edges, num_lines, seq_len, kind="parallel", seeds=(0,), num_points=50,
                    point_dim=2):
    """Batch dict with the reference's keys for `len(seeds)` synthetic wireframes.

    num_edges: int or sequence (one per wireframe).  Edge i of wireframe w is the straight segment
    between two U(-1,1)^2 endpoints (PRNG keyed by the wireframe seed), resampled to `num_points`
    points; edges are sorted by (x0, y0, x1, y1).
    """
    if point_dim != 2:
        raise ValueError("synthetic wireframes are 2-D line drawings")
    seeds = list(seeds)
    if isinstance(num_edges, int):
        num_edges = [num_edges] * len(seeds)
    n_wf = len(seeds)
    inp = np.zeros((n_wf, num_lines, num_points, point_dim), dtype=np.float32)
    mask = np.ones((n_wf, num_lines), dtype=bool)
    for w, (seed, n) in enumerate(zip(seeds, num_edges)):
        if n > num_lines:
            raise ValueError("wireframe with %d edges exceeds num_lines=%d" % (n, num_lines))
        g = np.random.default_rng([0x5EED, int(seed)])
        ends = g.uniform(-1.0, 1.0, size=(n, 4))
        order = np.lexsort((ends[:, 3], ends[:, 2], ends[:, 1], ends[:, 0]))
        ends = ends[order]
        for i in range(n):
            inp[w, i] = _segment_points(ends[i, 0:2], ends[i, 2:4], num_points)
        mask[w, :n] = False
    if kind == "parallel":
        label = np.zeros((n_wf, num_lines, seq_len), dtype=np.int64)
    else:
        label = np.zeros((n_wf, seq_len), dtype=np.int64)
    return {
        "input": torch.from_numpy(inp),
        "input_mask": torch.from_numpy(mask),
        "label": torch.from_numpy(label),
        "num_input": [int(n) for n in num_edges],
    }


def make_extra_mask(case, batch):
    """Seeded extra pointer mask for the golden cases that carry `extra_mask_seed`: bool [B, L] (True =
    forbidden edge) with B = N*F sequences (parallel) or N (seq2seq); ~20 % of the real edges masked."""
    g = np.random.default_rng([0xC0ED, int(case["extra_mask_seed"])])
    n_wf, L = batch["input"].shape[0], batch["input"].shape[1]
    B = n_wf * max(batch["num_input"]) if case["kind"] == "parallel" else n_wf
    return torch.from_numpy(g.random((B, L)) < 0.2)


def make_module_state(shapes, seed=0, gain=2.0):
    """Name-keyed synthetic parameters for an arbitrary sub-module (the sub-module-surface goldens):
    `shapes` maps parameter name -> shape (e.g. from `module.state_dict()`); matrices are xavier-uniform x `gain`,
    LayerNorm weights (`norm*.weight`) 1 + 0.3 N(0,1), every other vector 0.3 N(0,1); integer buffers
    (`position`) count from 0.  Depends on names, shapes and the seed only."""
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        g = _rng("module:" + name, seed)
        leaf = name.split(".")[-1]
        if leaf == "position":
            out[name] = torch.arange(shape[-1], dtype=torch.long).reshape(shape)
            continue
        if len(shape) >= 2:
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            w = g.uniform(-bound, bound, size=shape) * gain
        elif "norm" in name and leaf == "weight":
            w = 1.0 + 0.3 * g.standard_normal(shape)
        else:
            w = 0.3 * g.standard_normal(shape)
        out[name] = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
    return out


def make_named_tensor(name, shape, seed=0, scale=1.0):
    """Seeded N(0, scale) float32 tensor keyed by `name` (inputs of the sub-module-surface goldens)."""
    g = _rng("input:" + name, seed)
    return torch.from_numpy(np.ascontiguousarray(scale * g.standard_normal(tuple(shape)), dtype=np.float32))
