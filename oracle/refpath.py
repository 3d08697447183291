"""ORACLE (test infrastructure, NOT product code): CPU restatement of the reference decode path.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file.
The shipped path (`faceformer_amd`) never imports it and has no CPU fallback.

What it is: a functional, plain-PyTorch fp32 eager restatement of the reference's greedy pointer
decode, written over a flat `state_dict` (Appendix B of SURVEY.md) instead of nn.Modules, mirroring the
reference **op for op** -- including its redundant work (cross-attention K/V re-projected at every
step over the `repeat_interleave`d memory, whole prefix re-gathered every step, projection applied to
all prefix rows) -- so that (a) results are bit-identical to the reference imported in the survey
container and (b) timing it on the GPU box's host cores is "the reference's CPU path".

Pinning: `oracle/make_golden.py` (run in the build container, where /root/reference exists) checks
this file bit-for-bit against the imported reference model and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` re-checks this file against those vectors everywhere.

Reference anchors (file:line in /root/reference):
  embedding            faceformer/embedding.py:23-38, 106-108
  encoder layer (pre)  faceformer/transformer.py:164-176 ; encoder stack 70-83
  decoder layer (pre)  faceformer/transformer.py:235-256 ; decoder stack 95-124
  parallel greedy loop faceformer/models/model_para.py:181-241 ; select_next 173-179
  seq2seq greedy loop  faceformer/models/model.py:169-219   ; select_next 161-167
  mask fill value      faceformer/utils.py:16-20
The attention arithmetic itself lives in torch (`F.multi_head_attention_forward`, slow path with
`need_weights=True`) and is called here exactly as `nn.MultiheadAttention` does.
"""
import torch
import torch.nn.functional as F

LN_EPS = 1e-5


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], LN_EPS)


def _mha(query, key, value, sd, prefix, num_head, key_padding_mask=None, attn_mask=None):
    """nn.MultiheadAttention(...)(q, k, value=v, attn_mask=..., key_padding_mask=...)[0] in eval mode."""
    embed = query.shape[-1]
    if key_padding_mask is not None:
        key_padding_mask = F._canonical_mask(
            mask=key_padding_mask, mask_name="key_padding_mask",
            other_type=F._none_or_dtype(attn_mask), other_name="attn_mask",
            target_type=query.dtype)
    if attn_mask is not None:
        attn_mask = F._canonical_mask(
            mask=attn_mask, mask_name="attn_mask", other_type=None, other_name="",
            target_type=query.dtype, check_other=False)
    out, _ = F.multi_head_attention_forward(
        query, key, value, embed, num_head,
        sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"],
        None, None, False, 0.0,
        sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"],
        training=False, key_padding_mask=key_padding_mask, need_weights=True,
        attn_mask=attn_mask, average_attn_weights=True, is_causal=False)
    return out


def embed_edges(sd, coord, num_token):
    """reference embedding.py:23-38: token rows ++ MLP(flattened points)."""
    n = coord.size(0)
    token = torch.arange(num_token, dtype=torch.long)
    token_embed = F.embedding(token, sd["val_enc.embedding_token.weight"])
    token_embed = token_embed.unsqueeze(0).expand(n, num_token, -1)
    h = F.linear(coord.flatten(-2, -1), sd["val_enc.embedding_value.0.weight"],
                 sd["val_enc.embedding_value.0.bias"])
    h = F.relu(h)
    h = F.linear(h, sd["val_enc.embedding_value.2.weight"], sd["val_enc.embedding_value.2.bias"])
    return torch.cat((token_embed, h), dim=1)


def encoder_layer(sd, p, src, key_padding_mask, pos, num_head, act=F.relu, src_mask=None):
    """reference transformer.py:164-176 (forward_pre)."""
    y = _ln(src, sd, p + ".norm1")
    q = k = y + pos
    src = src + _mha(q, k, y, sd, p + ".self_attn", num_head, key_padding_mask, src_mask)
    y = _ln(src, sd, p + ".norm2")
    y = F.linear(act(F.linear(y, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return src + y


def encoder(sd, src, key_padding_mask, pos, num_head, num_layers):
    """reference transformer.py:70-83."""
    out = src
    for i in range(num_layers):
        out = encoder_layer(sd, "encoder.layers.%d" % i, out, key_padding_mask, pos, num_head)
    return _ln(out, sd, "encoder.norm")


def decoder_layer(sd, p, tgt, memory, memory_key_padding_mask, pos, query_pos, num_head,
                  tgt_mask=None):
    """reference transformer.py:235-256 (forward_pre)."""
    y = _ln(tgt, sd, p + ".norm1")
    q = k = y + query_pos
    tgt = tgt + _mha(q, k, y, sd, p + ".self_attn", num_head, None, tgt_mask)
    y = _ln(tgt, sd, p + ".norm2")
    tgt = tgt + _mha(y + query_pos, memory + pos, memory, sd, p + ".multihead_attn", num_head,
                     memory_key_padding_mask)
    y = _ln(tgt, sd, p + ".norm3")
    y = F.linear(F.relu(F.linear(y, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return tgt + y


def encoder_layer_post(sd, p, src, key_padding_mask, pos, num_head, act=F.relu, src_mask=None):
    """reference transformer.py:148-162 (forward_post; unused by the reference's configs, part of the module
    surface)."""
    q = k = src if pos is None else src + pos
    src = src + _mha(q, k, src, sd, p + ".self_attn", num_head, key_padding_mask, src_mask)
    src = _ln(src, sd, p + ".norm1")
    y = F.linear(act(F.linear(src, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return _ln(src + y, sd, p + ".norm2")


def decoder_layer_post(sd, p, tgt, memory, memory_key_padding_mask, pos, query_pos, num_head, tgt_mask=None,
                       tgt_key_padding_mask=None, act=F.relu, memory_mask=None):
    """reference transformer.py:211-233 (forward_post)."""
    q = k = tgt if query_pos is None else tgt + query_pos
    tgt = tgt + _mha(q, k, tgt, sd, p + ".self_attn", num_head, tgt_key_padding_mask, tgt_mask)
    tgt = _ln(tgt, sd, p + ".norm1")
    tgt = tgt + _mha(tgt if query_pos is None else tgt + query_pos, memory if pos is None else memory + pos, memory,
                     sd, p + ".multihead_attn", num_head, memory_key_padding_mask, memory_mask)
    tgt = _ln(tgt, sd, p + ".norm2")
    y = F.linear(act(F.linear(tgt, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return _ln(tgt + y, sd, p + ".norm3")


def decoder_layer_pre_kpm(sd, p, tgt, memory, memory_key_padding_mask, pos, query_pos, num_head, tgt_mask=None,
                          tgt_key_padding_mask=None, act=F.relu, memory_mask=None):
    """reference transformer.py:235-256 with EVERY keyword of the layer (the eval loop passes no tgt masks; the
    teacher-forced caller model_para.py:162-163 passes tgt_mask and tgt_key_padding_mask)."""
    y = _ln(tgt, sd, p + ".norm1")
    q = k = y if query_pos is None else y + query_pos
    tgt = tgt + _mha(q, k, y, sd, p + ".self_attn", num_head, tgt_key_padding_mask, tgt_mask)
    y = _ln(tgt, sd, p + ".norm2")
    tgt = tgt + _mha(y if query_pos is None else y + query_pos, memory if pos is None else memory + pos, memory, sd,
                     p + ".multihead_attn", num_head, memory_key_padding_mask, memory_mask)
    y = _ln(tgt, sd, p + ".norm3")
    y = F.linear(act(F.linear(y, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return tgt + y


def _activation(name):
    """reference transformer.py:276-284 (_get_activation_fn; "glu" halves the width and cannot feed linear2)."""
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    raise RuntimeError("activation should be relu/gelu, not %s." % name)


def decoder_stack(sd, prefix, tgt, memory, num_head, num_layers, normalize_before=True, final_norm=True,
                  return_intermediate=False, tgt_mask=None, tgt_key_padding_mask=None, memory_key_padding_mask=None,
                  pos=None, query_pos=None, activation="relu", memory_mask=None):
    """reference transformer.py:95-124 over a state_dict with keys `<prefix>layers.<i>.*`, `<prefix>norm.*`: both
    layer forms and the `return_intermediate` stack."""
    layer = decoder_layer_pre_kpm if normalize_before else decoder_layer_post
    out, inter = tgt, []
    for i in range(num_layers):
        out = layer(sd, "%slayers.%d" % (prefix, i), out, memory, memory_key_padding_mask, pos, query_pos, num_head,
                    tgt_mask, tgt_key_padding_mask, _activation(activation), memory_mask)
        if return_intermediate:
            inter.append(_ln(out, sd, prefix + "norm"))
    if final_norm:
        out = _ln(out, sd, prefix + "norm")
        if return_intermediate:
            inter[-1] = out
    return torch.stack(inter) if return_intermediate else out


def encoder_stack(sd, prefix, src, num_head, num_layers, normalize_before=True, final_norm=True,
                  src_key_padding_mask=None, pos=None, activation="relu", src_mask=None):
    """reference transformer.py:70-83 with either layer form."""
    out = src
    for i in range(num_layers):
        p = "%slayers.%d" % (prefix, i)
        if normalize_before:
            out = encoder_layer(sd, p, out, src_key_padding_mask, 0 if pos is None else pos, num_head, _activation(activation), src_mask)
        else:
            out = encoder_layer_post(sd, p, out, src_key_padding_mask, pos, num_head, _activation(activation), src_mask)
    return _ln(out, sd, prefix + "norm") if final_norm else out


def decoder(sd, tgt, memory, memory_key_padding_mask, pos, query_pos, num_head, num_layers,
            tgt_mask=None):
    """reference transformer.py:95-124 (return_intermediate=False)."""
    out = tgt
    for i in range(num_layers):
        out = decoder_layer(sd, "decoder.layers.%d" % i, out, memory, memory_key_padding_mask,
                            pos, query_pos, num_head, tgt_mask)
    return _ln(out, sd, "decoder.norm")


def select_next(memory, pointer, input_mask, extra_mask=None):
    """reference model_para.py:173-179 / model.py:161-167.  Returns (next_token 1xB, logit BxS)."""
    embedding = memory.transpose(0, 1)
    ptr = pointer.permute(1, 2, 0)
    logit = torch.bmm(embedding, ptr[..., -1:])
    fill = torch.finfo(logit.dtype).min
    logit = logit.masked_fill(input_mask.unsqueeze(-1), fill)
    if extra_mask is not None:  # optional co-edge style extra mask, default off
        logit = logit.masked_fill(extra_mask.unsqueeze(-1), fill)
    next_token = torch.argmax(logit, dim=1).transpose(0, 1)
    return next_token, logit.squeeze(-1)


def _stacks(sd, num_head, n_enc, n_dec, normalize_before, activation):
    """(encoder, decoder) callables of a model built with these constructor arguments.  The defaults of every reference config
    (pre-norm, relu) take the functions above; anything else the general stacks (post-norm: no `encoder.norm`, model.py:36)."""
    if normalize_before and activation == "relu":
        return (lambda src, mask, pos: encoder(sd, src, mask, pos, num_head, n_enc),
                lambda tgt, memory, mask, pos, qpos: decoder(sd, tgt, memory, mask, pos, qpos, num_head, n_dec))
    # The reference's weights are nn.Parameters (requires_grad=True, also under torch.no_grad()), and ATen's matmul folds a
    # NON-CONTIGUOUS 3-D input into one GEMM only when nothing requires grad -- otherwise it takes another GEMM shape, with
    # another summation order.  A post-norm encoder hands its first projections the transposed (non-contiguous) embedding
    # (model.py:186), so a bit-exact restatement has to present its weights the same way.  (The pre-norm path normalises
    # first: every projection input is contiguous there and the flag changes nothing.)
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    return (lambda src, mask, pos: encoder_stack(sd, "encoder.", src, num_head, n_enc, normalize_before,
                                                 final_norm=normalize_before, src_key_padding_mask=mask, pos=pos,
                                                 activation=activation),
            lambda tgt, memory, mask, pos, qpos: decoder_stack(sd, "decoder.", tgt, memory, num_head, n_dec, normalize_before,
                                                               final_norm=True, memory_key_padding_mask=mask, pos=pos,
                                                               query_pos=qpos, activation=activation))


def _dims(sd):
    num_model = sd["project.weight"].shape[0]
    n_enc = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    n_dec = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.layers."))
    num_token = sd["val_enc.embedding_token.weight"].shape[0]
    return num_model, n_enc, n_dec, num_token


@torch.no_grad()
def parallel_forward_eval(sd, inputs, num_head=8, max_face_length=None, trace=None,
                          anchor_limit=None, stop_rule=True, num_anchors=None, extra_mask=None,
                          normalize_before=True, activation="relu"):
    """SurfaceFormer_Parallel.forward_eval (reference model_para.py:181-241).

    `trace`: optional dict; receives 'logits' (list of BxS tensors per step) and 'memory'.
    `anchor_limit`: ONLY for the bounded cpu_baseline timing sample -- keep the first
    `anchor_limit` anchor sequences of every wireframe (sequences are independent, so their tokens
    are unchanged); `None` reproduces the reference exactly.
    `stop_rule=False` / `num_anchors`: ONLY for the multi-process tests -- run all T-1 steps and record
    the per-step special-token counts in trace['counts'] (a shard cannot evaluate the batch-global
    stop rule alone), and pad the anchor set to the batch-global F = max(num_input).
    `normalize_before` / `activation`: the constructor arguments the reference hands to its layers (model_para.py:14-19,
    33-45; no reference config changes them): post-norm layers (transformer.py:148-162, 211-233; `encoder.norm` is then
    None, model_para.py:36) and gelu feed-forward layers.
    """
    num_model, n_enc, n_dec, num_token = _dims(sd)
    run_encoder, run_decoder = _stacks(sd, num_head, n_enc, n_dec, normalize_before, activation)
    inp, input_mask, label = inputs["input"], inputs["input_mask"], inputs["label"]
    T = max_face_length if max_face_length is not None else sd["query_pos_enc.pos_embed.weight"].shape[0]
    batch_size = inp.size(0)
    max_num_edges = int(max(int(x) for x in inputs["num_input"]))
    if num_anchors is not None:
        max_num_edges = int(num_anchors)

    padding_mask = torch.zeros((len(input_mask), num_token)).type_as(input_mask)
    input_mask = torch.cat([padding_mask, input_mask], dim=1)

    val_embed = embed_edges(sd, inp, num_token)                                    # N x S x E
    pos_embed = sd["pos_enc.pos_embed.weight"][: val_embed.size(1)].unsqueeze(0)   # 1 x S x E
    qlen = label.transpose(1, 2).size(1)
    query_pos_embed = sd["query_pos_enc.pos_embed.weight"][:qlen].unsqueeze(0)     # 1 x T x E

    source, pos_embed = val_embed.transpose(0, 1), pos_embed.transpose(0, 1)

    anchors = torch.arange(max_num_edges).repeat(1, batch_size, 1).type_as(label)
    for i, num_edges in enumerate(inputs["num_input"]):
        anchors[:, i, int(num_edges):] = num_token - 1
    if anchor_limit is not None:
        anchors = anchors[:, :, :anchor_limit]
        max_num_edges = anchors.size(2)
    query_pos_embed = query_pos_embed.transpose(0, 1)
    predicts = anchors.flatten(1, 2)

    memory = run_encoder(source, input_mask, pos_embed)
    if trace is not None:
        trace["memory"] = memory.transpose(0, 1).clone()
        trace["logits"] = []
        trace["counts"] = []
    memory = memory.repeat_interleave(max_num_edges, 1)
    input_mask = input_mask.repeat_interleave(max_num_edges, 0)
    if extra_mask is not None:  # [N*F, L] bool, per sequence; special-token columns are never masked
        extra_mask = torch.cat([torch.zeros((extra_mask.size(0), num_token)).type_as(extra_mask), extra_mask], dim=1)

    for step in range(T - 1):
        target = predicts.unsqueeze(-1).repeat(1, 1, num_model)
        tgt = torch.gather(memory, 0, target)
        pointer = run_decoder(tgt, memory, input_mask, pos_embed, query_pos_embed[: step + 1])
        pointer = F.linear(pointer, sd["project.weight"], sd["project.bias"])
        next_token, logit = select_next(memory, pointer, input_mask, extra_mask)
        if trace is not None:
            trace["logits"].append(logit.clone())
        predicts = torch.cat((predicts, next_token), dim=0)
        if trace is not None:
            trace["counts"].append(int((next_token >= num_token).sum()))
        if stop_rule and torch.all(next_token < num_token):
            break

    predicts = torch.cat(
        (predicts, torch.zeros(T - predicts.size(0), predicts.size(1)).type_as(predicts)), dim=0)
    inputs["predict"] = predicts.transpose(0, 1).view(-1, max_num_edges, T)
    return inputs


@torch.no_grad()
def seq2seq_forward_eval(sd, inputs, num_head=8, label_seq_length=None, token_sos=1, token_eos=3,
                         trace=None, extra_mask=None, normalize_before=True, activation="relu"):
    """SurfaceFormer.forward_eval (reference model.py:169-219); `normalize_before` / `activation` as above (model.py:14-18)."""
    num_model, n_enc, n_dec, num_token = _dims(sd)
    run_encoder, run_decoder = _stacks(sd, num_head, n_enc, n_dec, normalize_before, activation)
    inp, input_mask, label = inputs["input"], inputs["input_mask"], inputs["label"]
    T = label_seq_length if label_seq_length is not None else sd["query_pos_enc.pos_embed.weight"].shape[0]
    batch_size = inp.size(0)

    padding_mask = torch.zeros((len(input_mask), num_token)).type_as(input_mask)
    input_mask = torch.cat([padding_mask, input_mask], dim=1)
    if extra_mask is not None:
        extra_mask = torch.cat([padding_mask, extra_mask], dim=1)

    val_embed = embed_edges(sd, inp, num_token)
    pos_embed = sd["pos_enc.pos_embed.weight"][: val_embed.size(1)].unsqueeze(0)
    query_pos_embed = sd["query_pos_enc.pos_embed.weight"][: label.size(1)].unsqueeze(0)

    source, pos_embed = val_embed.transpose(0, 1), pos_embed.transpose(0, 1)
    query_pos_embed = query_pos_embed.transpose(0, 1)

    memory = run_encoder(source, input_mask, pos_embed)
    if trace is not None:
        trace["memory"] = memory.transpose(0, 1).clone()
        trace["logits"] = []

    predicts = torch.full((1, batch_size), token_sos, dtype=torch.long)
    eos_found = 0
    pointer = None
    for step in range(T - 1):
        target = predicts.unsqueeze(-1).repeat(1, 1, num_model)
        tgt = torch.gather(memory, 0, target)
        pointer = run_decoder(tgt, memory, input_mask, pos_embed, query_pos_embed[: step + 1])
        pointer = F.linear(pointer, sd["project.weight"], sd["project.bias"])
        next_token, logit = select_next(memory, pointer, input_mask, extra_mask)
        if trace is not None:
            trace["logits"].append(logit.clone())
        predicts = torch.cat((predicts, next_token), dim=0)
        eos_found += next_token.eq(token_eos).sum().item()
        if eos_found == batch_size:
            break

    predicts = torch.cat(
        (predicts, torch.zeros(T - predicts.size(0), predicts.size(1)).type_as(predicts)), dim=0)
    inputs["embedding"] = memory.transpose(0, 1)
    inputs["pointer"] = pointer.transpose(0, 1)
    inputs["predict"] = predicts.transpose(0, 1)
    return inputs
