"""Transformer blocks of the decode path on hand-written HIP kernels (gfx950).

Module surface, constructor signatures, parameter names / shapes and forward keyword arguments follow
reference `faceformer/transformer.py` (encoder stack 62-83, decoder stack 86-124, encoder layer
127-184, decoder layer 187-269, helpers 272-284) so `state_dict`s are interchangeable.  The
arithmetic is NOT torch's: every forward below runs on libfaceformer_hip.so --

    LayerNorm (+pos add)      -> ff_layernorm      (one wavefront per row)
    q/k/v, out-proj, FFN      -> ff_gemm_f32       (v_mfma_f32_32x32x2_f32, fused bias/ReLU/residual)
    softmax(qk^T)v            -> ff_attention      (LDS-staged K/V tiles, online softmax, f32 MFMA; 64-wide heads,
                                                    key padding / causal masks: every reference config)
                                 ff_attention_general (any num_model / num_head, torch's general `attn_mask` forms:
                                                    src_mask, non-causal / float tgt_mask, memory_mask)

Tensors are sequence-first (len x batch x E) like the reference; a contiguous sequence-first tensor
IS the position-major row matrix the kernels want (row = position * batch + b), so no transposes
happen.  Inference only: a module in training mode raises (training is out of scope of this build),
and CPU tensors raise (there is no fallback path).
"""
import copy
from typing import Optional

import torch
from torch import Tensor, nn

from .hip import ops
from .hip.lib import FF_HEAD_DIM, HipExtensionError

__all__ = ["Transformer", "TransformerEncoder", "TransformerDecoder", "TransformerEncoderLayer",
           "TransformerDecoderLayer", "MultiheadAttention", "HipLinear"]


def _eval_only(mod):
    if mod.training:
        raise NotImplementedError(
            "%s: training-mode forward (dropout / teacher forcing) is out of scope of the MI355X "
            "decode build; call .eval()" % type(mod).__name__)


def _rows(x):
    """[len, batch, E] -> contiguous [len*batch, E] row matrix (a view when already contiguous)."""
    if x.dim() != 3:
        raise ValueError("expected a sequence-first [len, batch, E] tensor, got %s" % (tuple(x.shape),))
    return x.contiguous().view(x.size(0) * x.size(1), x.size(2))


def _pos_table(pos, length, batch):
    """Positional term as (table, pos_div, pos_mod) for the kernels' `(row // div) % mod` lookup.
    Accepts [len,1,E] (broadcast over batch, what the path uses) or a full [len,batch,E]."""
    if pos is None:
        return None, 1, 1
    if pos.dim() != 3 or pos.size(0) != length:
        raise ValueError("positional term must be [len, 1|batch, E] with len=%d" % length)
    if pos.size(1) == 1:
        return pos.contiguous().view(length, pos.size(2)), batch, length
    if pos.size(1) == batch:
        return pos.contiguous().view(length * batch, pos.size(2)), 1, length * batch
    raise ValueError("positional term batch dim must be 1 or %d" % batch)


def _mask_u8(mask, name, shape):
    if mask is None:
        return None
    if tuple(mask.shape) != tuple(shape):
        raise ValueError("%s must have shape %s, got %s" % (name, shape, tuple(mask.shape)))
    if mask.is_floating_point():    # torch adds a floating-point padding mask to the scores of its keys (attend -> ff_attention_general)
        return mask.to(torch.float32).contiguous()
    if mask.dtype not in (torch.bool, torch.uint8):
        raise TypeError("%s must be a boolean or floating-point tensor, got %s" % (name, mask.dtype))
    return mask.to(torch.uint8).contiguous()


def _attn_mask_forms(mask, name, lq, lk, batch_heads):
    """torch's `attn_mask` (nn.MultiheadAttention: 2-D [L, S] or 3-D [N*H, L, S]; boolean = True removes the key, floating =
    added to the scores) -> (causal, attn_bias, attn_mask) for the kernels.  The square 'subsequent' boolean mask (reference
    model.py:71-73) is recognised and runs as the MFMA kernels' causal rule; every other form goes to ff_attention_general."""
    if mask is None:
        return False, None, None
    if mask.dim() == 2:
        want = (lq, lk)
    elif mask.dim() == 3:
        want = (batch_heads, lq, lk)
    else:
        raise ValueError("%s must be 2-D [L, S] or 3-D [N*num_heads, L, S], got %s" % (name, tuple(mask.shape)))
    if tuple(mask.shape) != want:
        raise ValueError("%s must have shape %s, got %s" % (name, want, tuple(mask.shape)))
    if mask.dtype in (torch.bool, torch.uint8):
        m = mask.to(torch.bool)
        if m.dim() == 2 and lq == lk and torch.equal(
                m, torch.triu(torch.ones(lq, lq, dtype=torch.bool, device=m.device), diagonal=1)):
            return True, None, None
        return False, None, m.to(torch.uint8).contiguous()
    if mask.is_floating_point():
        return False, mask.to(torch.float32).contiguous(), None
    raise TypeError("%s must be a boolean or floating-point tensor, got %s" % (name, mask.dtype))


class MultiheadAttention(nn.Module):
    """Parameter container with torch's nn.MultiheadAttention names/shapes (`in_proj_weight` [3E,E],
    `in_proj_bias`, `out_proj.{weight,bias}`) whose forward runs on the HIP kernels.
    forward(query, key, value, key_padding_mask=None, attn_mask=None) -> (output, None)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _check(self):
        _eval_only(self)

    def _two_inputs(self, xa, xb, w, b, n_split):
        """columns [0, n_split) from `xa`, the rest from `xb`, one launch -- the kernel switches its A operand at a 64-column tile
        boundary, so widths that are not a multiple of 64 (num_model 96, ...) take two launches into one output buffer."""
        if xa is xb:
            return ops.linear(xa, w, b)
        if n_split % 64 == 0:
            return ops.linear(xa, w, b, x2=xb, n_split=n_split)
        out = torch.empty((xa.size(0), w.size(0)), device=xa.device, dtype=torch.float32)
        ops.linear(xa, w[:n_split], b[:n_split], out=out[:, :n_split])
        ops.linear(xb, w[n_split:], b[n_split:], out=out[:, n_split:])
        return out

    def project_self(self, yq, y):
        """q|k from `yq`, v from `y` in one launch -> [rows, 3E]."""
        return self._two_inputs(yq, y, self.in_proj_weight, self.in_proj_bias, 2 * self.embed_dim)

    def project_q(self, yq):
        E = self.embed_dim
        return ops.linear(yq, self.in_proj_weight[:E], self.in_proj_bias[:E])

    def project_kv(self, kin, vin):
        """k from `kin`, v from `vin` -> [rows, 2E]."""
        E = self.embed_dim
        return self._two_inputs(kin, vin, self.in_proj_weight[E:], self.in_proj_bias[E:], E)

    def attend(self, q, k, v, lq, lk, batch, key_padding_mask=None, causal=False, attn_bias=None, attn_mask=None):
        """q: [lq*batch, >=E] rows (position-major), k/v: [lk*batch, ...]; returns [lq*batch, E].  64-wide heads with a key
        padding mask and / or the causal rule run on the MFMA kernels (ff_attention); any other head width and torch's general
        `attn_mask` forms (attn_bias additive, attn_mask boolean: _attn_mask_forms) run on ff_attention_general."""
        kv_len = None
        if key_padding_mask is not None and key_padding_mask.is_floating_point():
            # additive padding mask [batch, lk]: one [lq, lk] matrix per (batch, head) like torch's merged mask (mask plumbing only)
            pad = key_padding_mask[:, None, None, :].expand(batch, self.num_heads, lq, lk)
            if attn_bias is not None:
                pad = pad + (attn_bias.view(batch, self.num_heads, lq, lk) if attn_bias.dim() == 3 else attn_bias)
            if attn_mask is not None and attn_mask.dim() == 2:
                attn_mask = attn_mask[None].expand(batch * self.num_heads, lq, lk)
            attn_bias, key_padding_mask = pad.reshape(batch * self.num_heads, lq, lk).contiguous(), None
        elif attn_bias is not None and attn_mask is not None and attn_bias.dim() != attn_mask.dim():
            if attn_bias.dim() == 2:
                attn_bias = attn_bias[None].expand(batch * self.num_heads, lq, lk).contiguous()
            else:
                attn_mask = attn_mask[None].expand(batch * self.num_heads, lq, lk).contiguous()
        if key_padding_mask is not None:
            idx = torch.arange(1, lk + 1, device=q.device, dtype=torch.int32)
            kv_len = ((key_padding_mask == 0).to(torch.int32) * idx).amax(dim=1).to(torch.int32)
        if self.head_dim != FF_HEAD_DIM or attn_bias is not None or attn_mask is not None:
            return ops.attention_general(q, k, v, num_groups=batch, num_heads=self.num_heads, head_dim=self.head_dim, nq=lq, nk=lk,
                                         q_group_stride=1, q_inner=1, q_outer_stride=batch,
                                         k_group_stride=1, k_stride=batch, kv_len=kv_len, key_mask=key_padding_mask,
                                         causal=causal, attn_bias=attn_bias, attn_mask=attn_mask,
                                         scale=float(self.head_dim) ** -0.5)
        return ops.attention(q, k, v, num_groups=batch, num_heads=self.num_heads, nq=lq, nk=lk,
                             q_group_stride=1, q_inner=1, q_outer_stride=batch,
                             k_group_stride=1, k_stride=batch, kv_len=kv_len,
                             key_mask=key_padding_mask, causal=causal,
                             scale=float(self.head_dim) ** -0.5)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=False, attn_mask=None):
        self._check()
        lq, batch, E = query.shape
        lk = key.size(0)
        q = self.project_q(_rows(query))
        kv = self.project_kv(_rows(key), _rows(value))
        kpm = _mask_u8(key_padding_mask, "key_padding_mask", (batch, lk))
        causal, bias, amask = _attn_mask_forms(attn_mask, "attn_mask", lq, lk, batch * self.num_heads)
        o = self.attend(q, kv[:, :E], kv[:, E:], lq, lk, batch, kpm, causal, bias, amask)
        out = ops.linear(o, self.out_proj.weight, self.out_proj.bias)
        return out.view(lq, batch, E), None


class HipLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose forward runs on the f32-MFMA projection kernel: the
    models' `project` head (reference model_para.py:46,225) when an external caller drives the blocks itself."""

    def forward(self, x):
        _eval_only(self)
        out = ops.linear(x.reshape(-1, x.size(-1)), self.weight, self.bias)
        return out.view(*x.shape[:-1], self.out_features)


def _activation_code(name):
    """1: ReLU in the projection's epilogue; 2: exact GELU as a row op behind the plain projection (module surface only: the
    reference configs use relu, config.py / model.py:16)."""
    if name == "relu":
        return 1
    if name == "gelu":
        return 2
    if name == "glu":
        # F.glu halves the feature dimension, so the reference's own linear2 (num_feedforward inputs) cannot consume it:
        # transformer.py:283 accepts the name, the forward fails in torch.  Say so up front.
        raise NotImplementedError("activation 'glu' halves the feed-forward width: the reference's linear2 cannot consume it "
                                  "either (transformer.py:283 accepts the name, its forward raises)")
    raise RuntimeError("activation should be relu/gelu, not %s." % name)


def _get_activation_fn(activation):
    """Kept for API parity (reference transformer.py:276-284); returns the torch functional."""
    import torch.nn.functional as F
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError("activation should be relu/gelu, not %s." % activation)


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class TransformerEncoderLayer(nn.Module):
    """Reference transformer.py:127-184."""

    def __init__(self, num_model, num_head, num_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadAttention(num_model, num_head, dropout=dropout)
        self.linear1 = nn.Linear(num_model, num_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(num_feedforward, num_model)
        self.norm1 = nn.LayerNorm(num_model)
        self.norm2 = nn.LayerNorm(num_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation_name = activation
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    def with_pos_embed(self, tensor, pos: Optional[Tensor]):
        if pos is None:
            return tensor
        table, div, mod = _pos_table(pos, tensor.size(0), tensor.size(1))
        return ops.add_pos(_rows(tensor), table, div, mod).view_as(tensor)

    def _ffn(self, y, residual):
        act = _activation_code(self.activation_name)
        h = ops.linear(y, self.linear1.weight, self.linear1.bias, act=1 if act == 1 else 0)
        if act == 2:
            ops.gelu_(h)
        return ops.linear(h, self.linear2.weight, self.linear2.bias, residual=residual)

    def _prep(self, src, src_mask, src_key_padding_mask, pos):
        self.self_attn._check()
        S, N, E = src.shape
        kpm = _mask_u8(src_key_padding_mask, "src_key_padding_mask", (N, S))
        smask = _attn_mask_forms(src_mask, "src_mask", S, S, N * self.self_attn.num_heads)
        return S, N, E, kpm, smask, _pos_table(pos, S, N)

    def forward_pre(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        S, N, E, kpm, smask, (table, div, mod) = self._prep(src, src_mask, src_key_padding_mask, pos)
        x = _rows(src)
        y, yq = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, table, div, mod)
        qkv = self.self_attn.project_self(yq if yq is not None else y, y)
        o = self.self_attn.attend(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], S, S, N, kpm, *smask)
        x = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, residual=x)
        y, _ = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self._ffn(y, x).view(S, N, E)

    def forward_post(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        S, N, E, kpm, smask, (table, div, mod) = self._prep(src, src_mask, src_key_padding_mask, pos)
        x = _rows(src)
        xq = ops.add_pos(x, table, div, mod) if table is not None else x
        qkv = self.self_attn.project_self(xq, x)
        o = self.self_attn.attend(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], S, S, N, kpm, *smask)
        x = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, residual=x)
        x, _ = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        x = self._ffn(x, x)
        x, _ = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return x.view(S, N, E)

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        _eval_only(self)
        if self.normalize_before:
            return self.forward_pre(src, src_mask, src_key_padding_mask, pos)
        return self.forward_post(src, src_mask, src_key_padding_mask, pos)


class TransformerDecoderLayer(nn.Module):
    """Reference transformer.py:187-269."""

    def __init__(self, num_model, num_head, num_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = MultiheadAttention(num_model, num_head, dropout=dropout)
        self.multihead_attn = MultiheadAttention(num_model, num_head, dropout=dropout)
        self.linear1 = nn.Linear(num_model, num_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(num_feedforward, num_model)
        self.norm1 = nn.LayerNorm(num_model)
        self.norm2 = nn.LayerNorm(num_model)
        self.norm3 = nn.LayerNorm(num_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.activation_name = activation
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before

    with_pos_embed = TransformerEncoderLayer.with_pos_embed
    _ffn = TransformerEncoderLayer._ffn

    def _prep(self, tgt, memory, tgt_mask, memory_mask, tgt_kpm, mem_kpm, pos, query_pos):
        self.self_attn._check()
        t, B, E = tgt.shape
        S = memory.size(0)
        if memory.size(1) != B:
            raise ValueError("memory batch %d != tgt batch %d" % (memory.size(1), B))
        return (t, B, E, S, _attn_mask_forms(tgt_mask, "tgt_mask", t, t, B * self.self_attn.num_heads),
                _attn_mask_forms(memory_mask, "memory_mask", t, S, B * self.multihead_attn.num_heads),
                _mask_u8(tgt_kpm, "tgt_key_padding_mask", (B, t)),
                _mask_u8(mem_kpm, "memory_key_padding_mask", (B, S)),
                _pos_table(pos, S, B), _pos_table(query_pos, t, B))

    def _cross_kv(self, memory, ptab):
        mem = _rows(memory)
        table, div, mod = ptab
        kin = ops.add_pos(mem, table, div, mod) if table is not None else mem
        return self.multihead_attn.project_kv(kin, mem)

    def forward_pre(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                    memory_key_padding_mask=None, pos=None, query_pos=None):
        t, B, E, S, tmask, mmask, tkpm, mkpm, ptab, (qtab, qdiv, qmod) = self._prep(
            tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos, query_pos)
        x = _rows(tgt)
        y, yq = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, qtab, qdiv, qmod)
        qkv = self.self_attn.project_self(yq if yq is not None else y, y)
        o = self.self_attn.attend(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], t, t, B, tkpm, *tmask)
        x = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, residual=x)
        y, yq = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, qtab, qdiv, qmod)
        q = self.multihead_attn.project_q(yq if yq is not None else y)
        kv = self._cross_kv(memory, ptab)
        o = self.multihead_attn.attend(q, kv[:, :E], kv[:, E:], t, S, B, mkpm, *mmask)
        x = ops.linear(o, self.multihead_attn.out_proj.weight, self.multihead_attn.out_proj.bias, residual=x)
        y, _ = ops.layernorm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return self._ffn(y, x).view(t, B, E)

    def forward_post(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                     memory_key_padding_mask=None, pos=None, query_pos=None):
        t, B, E, S, tmask, mmask, tkpm, mkpm, ptab, (qtab, qdiv, qmod) = self._prep(
            tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask, pos, query_pos)
        x = _rows(tgt)
        xq = ops.add_pos(x, qtab, qdiv, qmod) if qtab is not None else x
        qkv = self.self_attn.project_self(xq, x)
        o = self.self_attn.attend(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], t, t, B, tkpm, *tmask)
        x = ops.linear(o, self.self_attn.out_proj.weight, self.self_attn.out_proj.bias, residual=x)
        x, xq = ops.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, qtab, qdiv, qmod)
        q = self.multihead_attn.project_q(xq if xq is not None else x)
        kv = self._cross_kv(memory, ptab)
        o = self.multihead_attn.attend(q, kv[:, :E], kv[:, E:], t, S, B, mkpm, *mmask)
        x = ops.linear(o, self.multihead_attn.out_proj.weight, self.multihead_attn.out_proj.bias, residual=x)
        x, _ = ops.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        x = self._ffn(x, x)
        x, _ = ops.layernorm(x, self.norm3.weight, self.norm3.bias, self.norm3.eps)
        return x.view(t, B, E)

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        _eval_only(self)
        fn = self.forward_pre if self.normalize_before else self.forward_post
        return fn(tgt, memory, tgt_mask, memory_mask, tgt_key_padding_mask, memory_key_padding_mask,
                  pos, query_pos)


def _final_norm(norm, x):
    L, B, E = x.shape
    y, _ = ops.layernorm(_rows(x), norm.weight, norm.bias, norm.eps)
    return y.view(L, B, E)


class TransformerEncoder(nn.Module):
    """Reference transformer.py:62-83."""

    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None):
        _eval_only(self)
        output = src
        for layer in self.layers:
            output = layer(output, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos)
        if self.norm is not None:
            output = _final_norm(self.norm, output)
        return output


class TransformerDecoder(nn.Module):
    """Reference transformer.py:86-124."""

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        _eval_only(self)
        output = tgt
        intermediate = []
        for layer in self.layers:
            output = layer(output, memory, tgt_mask=tgt_mask, memory_mask=memory_mask,
                           tgt_key_padding_mask=tgt_key_padding_mask,
                           memory_key_padding_mask=memory_key_padding_mask, pos=pos, query_pos=query_pos)
            if self.return_intermediate:
                intermediate.append(_final_norm(self.norm, output))
        if self.norm is not None:
            output = _final_norm(self.norm, output)
            if self.return_intermediate:
                intermediate[-1] = output
        if self.return_intermediate:
            return torch.stack(intermediate)
        return output


class Transformer(nn.Module):
    """Encoder/decoder pair over an image-shaped feature map.  The reference carries such a wrapper
    (transformer.py:18-59) that neither model class instantiates; it exists here only so that
    `faceformer.transformer.Transformer(...)` keeps its constructor arguments and call contract:
    (src N x C x H x W, mask N x H x W, query_embed Q x C, pos_embed N x C x H x W) ->
    (decoder states transposed like the reference does, memory N x C x H x W)."""

    def __init__(self, num_model=512, num_head=8, num_encoder_layers=6, num_decoder_layers=6,
                 num_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False,
                 return_intermediate_dec=False):
        super().__init__()
        self.num_model, self.num_head = num_model, num_head
        layer_args = (num_model, num_head, num_feedforward, dropout, activation, normalize_before)
        self.encoder = TransformerEncoder(TransformerEncoderLayer(*layer_args), num_encoder_layers,
                                          norm=nn.LayerNorm(num_model) if normalize_before else None)
        self.decoder = TransformerDecoder(TransformerDecoderLayer(*layer_args), num_decoder_layers,
                                          norm=nn.LayerNorm(num_model), return_intermediate=return_intermediate_dec)
        for weight in (p for p in self.parameters() if p.dim() > 1):
            nn.init.xavier_uniform_(weight)

    @staticmethod
    def _as_tokens(maps):
        """N x C x H x W -> (H*W) x N x C (the sequence-first layout of the blocks above)."""
        return maps.reshape(maps.size(0), maps.size(1), -1).permute(2, 0, 1)

    def forward(self, src, mask, query_embed, pos_embed):
        n, c, h, w = src.shape
        pos, key_pad = self._as_tokens(pos_embed), mask.reshape(n, -1)
        queries = query_embed[:, None, :].expand(-1, n, -1).contiguous()
        memory = self.encoder(self._as_tokens(src), src_key_padding_mask=key_pad, pos=pos)
        states = self.decoder(queries.new_zeros(queries.shape), memory, memory_key_padding_mask=key_pad,
                              pos=pos, query_pos=queries)
        # (the reference transposes dims 1 and 2 of WHATEVER the decoder returns, transformer.py:59: [layers, Q, N, C] ->
        #  [layers, N, Q, C] with return_intermediate_dec, [Q, N, C] -> [Q, C, N] without; kept as is)
        return states.transpose(1, 2), memory.permute(1, 2, 0).reshape(n, c, h, w)
