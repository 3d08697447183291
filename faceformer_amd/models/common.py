"""Shared plumbing of the two model classes: parameter containers with the reference's names, lazy
binding of those parameters into the native engine, mask/embedding helpers."""
import torch
import torch.nn as nn

from ..embedding import PositionEmbeddingLearned, VanillaEmedding
from ..hip import lib as _L
from ..hip import ops
from ..hip.engine import DEFAULT_FLAGS, PathEngine
from ..transformer import (HipLinear, TransformerDecoder, TransformerDecoderLayer, TransformerEncoder,
                           TransformerEncoderLayer)
from ..utils import min_value_of_dtype


SPLIT_KIND_DEFAULT = "fp16x2"   # round 6 (profiles/r06/split_kinds_ab.txt: config B 50.9 -> 45.4 ms, C128 256 k -> 302 k edges/s, margins unchanged)
X3_MIN_ROWS_DEFAULT = 1024   # package default of SurfaceFormerBase.x3_min_rows (bench.py reports this form beside the f32 headline)


class SurfaceFormerBase(nn.Module):
    """Everything SurfaceFormer and SurfaceFormer_Parallel have in common (reference
    models/model.py:14-69 and models/model_para.py:14-70 build identical sub-modules)."""

    def _build(self, num_model, num_head, num_feedforward, num_encoder_layers, num_decoder_layers,
               dropout, activation, normalize_before, num_points_per_line, num_lines, point_dim,
               seq_len, token, teacher_forcing_ratio):
        if token is None:
            raise ValueError("`token` (cfg.model.token) is required")
        self.num_model = num_model
        self.num_head = num_head
        self.teacher_forcing_ratio = teacher_forcing_ratio
        self.token = token
        self.num_token = token.len
        self.normalize_before = normalize_before
        self.activation_name = activation

        self.val_enc = VanillaEmedding(num_points_per_line * point_dim, num_model, token)
        self.pos_enc = PositionEmbeddingLearned(num_model, max_len=num_lines + self.num_token)
        self.query_pos_enc = PositionEmbeddingLearned(num_model, max_len=seq_len)

        enc_layer = TransformerEncoderLayer(num_model, num_head, num_feedforward, dropout, activation,
                                            normalize_before)
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers,
                                          nn.LayerNorm(num_model) if normalize_before else None)
        dec_layer = TransformerDecoderLayer(num_model, num_head, num_feedforward, dropout, activation,
                                            normalize_before)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, nn.LayerNorm(num_model))
        self.project = HipLinear(num_model, num_model)
        self._reset_parameters()

        # engine knobs (not part of the reference surface)
        self.decode_flags = DEFAULT_FLAGS
        self.chunk_wireframes = 16     # micro-batch size in wireframes (0 = whole batch); 8-32 measured best
        self.chunk_seqs = 0            # >0: split every wireframe into groups of this many sequences
        self.num_streams = 1           # micro-batches are issued round-robin on this many HIP streams
        self.chunk_max_seqs = 8192     # ... and at most this many sequences per micro-batch of several wireframes
        self.sort_by_edges = True      # ragged batches: decode the wireframes sorted by edge count (tight micro-batches)
        self.ln_fuse_max_rows = 0      # LayerNorm folded into the projections on steps with at most this many rows (0: 12288)
        self.sync_every = 1            # the host looks at the stop rule every k steps, one period behind the enqueued steps
                                       # (the queue never drains: free at every step, tools/run_sync_probe.sh); a decode
                                       # that stops at step s executes s + k ... s + 2k - 1 steps
        self.sharded_sync_every = 2    # ... period of the batch-global rule in dist.decode_sharded (a host all-reduce per check)
        # Decoder projections of launches with at least this many rows (q|k|v; linear1 from 7/4 x, the 512-column ones from
        # 11/4 x as many) run as 3 x bf16 split products on the bf16 matrix cores: fp32-accurate (error vs fp64 = an fp32 dot
        # product's, tests/test_hip_ops.py), LayerNorm folding included (ff_gemm_x3_ln), and 1.3-1.6x the f32-MFMA kernel
        # from ~3000 rows on (profiles/r04/gemm_x3_variants.txt).  0 = off.
        self.x3_min_rows = X3_MIN_ROWS_DEFAULT
        # ... their LayerNorm applied in the consumer's EPILOGUE (rstd (x W'^T - mean colsum(W'))): the K loop then is the plain
        # split product.  Error factor (1 + |mean| / sigma) of the row -- 0.06 median, 0.15 maximum on this model's LayerNorm
        # inputs; False = rows normalised before the product (ff_gemm_x3.hip: x3_ln_linear).  Read when the engine is bound.
        # None = by split kind: True for the bf16 terms (+5-13 % there), False for the fp16 terms (config B 44.3-44.8 vs 44.9 ms, C128 313 k vs
        # 309 k edges/s with the rows normalised first: profiles/r06/fp16x2_ln_form_ab.txt -- and normalised rows are bounded by sqrt(E),
        # so nothing un-normalised ever meets fp16's range).
        self.x3_ln_in_epilogue = None
        # how those projections split an fp32 operand: "bf16x3" (three bf16 terms, six products) or "fp16x2" (two fp16 terms, three
        # products: half the matrix-core work; the engine checks at bind time that the model's operand bounds fit fp16's range)
        self.split_kind = SPLIT_KIND_DEFAULT
        self._engine_obj = None

    def _reset_parameters(self):
        # every tensor with more than one dim is re-drawn xavier-uniform (reference model.py:49-52)
        for _, param in self.named_parameters():
            if param.dim() > 1:
                nn.init.xavier_uniform_(param)

    # ---- helpers that exist on the reference classes -------------------------------------------
    def process_masks(self, input_mask, tgt_mask=None):
        """Prepend `num_token` never-masked columns (reference model.py:61-69)."""
        pad = torch.zeros((len(input_mask), self.num_token), device=input_mask.device).type_as(input_mask)
        input_mask = torch.cat([pad, input_mask], dim=1)
        if tgt_mask is None:
            return input_mask
        return input_mask, tgt_mask[..., :-1].contiguous()

    def generate_square_subsequent_mask(self, sz):
        return torch.triu(torch.ones(sz, sz, dtype=torch.bool), diagonal=1)

    def patch_source(self, src, pos):
        return src.transpose(0, 1), pos.transpose(0, 1)

    def select_next(self, embedding, pointer, input_mask):
        """embedding S x B x E, pointer t x B x E, input_mask B x S (True = masked) -> 1 x B tokens
        (reference model.py:161-167); runs the HIP pointer kernel."""
        mem = embedding.transpose(0, 1).contiguous()
        res = ops.pointer_argmax(pointer[-1].contiguous(), mem, mask=input_mask.to(torch.uint8).contiguous(),
                                 seqs_per_group=1)
        return res["next"].to(torch.long).unsqueeze(0)

    def forward_train(self, inputs, scheduled_sampling_ratio=0):
        raise NotImplementedError(
            "teacher-forced training (reference forward_train) is out of scope of the MI355X decode "
            "build; this package implements the greedy eval path")

    # ---- native engine binding --------------------------------------------------------------------
    def engine_supported(self):
        """The native engine (ff_encode / ff_decode) implements what every reference config uses: pre-norm layers with relu
        feed-forward blocks and 64-wide attention heads (num_model / num_head = 512 / 8).  The other constructor arguments of
        the reference (model.py:14-18: `normalize_before=False`, `activation="gelu"`, any head width) decode through
        `_forward_eval_modules`."""
        return bool(self.normalize_before) and self.activation_name == "relu" and \
            self.num_model == self.num_head * _L.FF_HEAD_DIM

    def _check_supported(self):
        if not self.normalize_before:
            raise NotImplementedError("the native decode engine implements the pre-norm path "
                                      "(normalize_before=True, what every reference config uses)")
        if self.activation_name != "relu":
            raise NotImplementedError("the native decode engine implements relu feed-forward layers")
        if self.num_model != self.num_head * _L.FF_HEAD_DIM:
            raise NotImplementedError("the native decode engine implements %d-wide attention heads (num_model / num_head = 512 / 8 in "
                                      "every reference config); other widths decode through the sub-module loop" % _L.FF_HEAD_DIM)

    def _forward_eval_modules(self, inputs, parallel):
        """Greedy decode of a model the native engine does not implement (post-norm layers, gelu, head widths other than 64), driven from Python over
        this package's HIP sub-modules: `encoder` / `decoder` / `project` run ff_layernorm / ff_gemm_f32 / ff_attention /
        ff_gelu, the pointer head is ff_pointer_argmax.  Same semantics as the reference's loops (model_para.py:181-241,
        model.py:169-219): whole prefix re-decoded every step, `finfo.min` mask fill, lowest index on ties, the parallel
        model's current-step stop rule / the single-sequence model's cumulative EOS count, zero padding afterwards.
        One launch per operator and a host synchronisation per step (the stop rule): a module-level path for constructor
        arguments no reference config uses, not the tuned engine -- still HIP kernels only, no CPU fallback."""
        inp, label = inputs["input"], inputs["label"]
        if not inp.is_cuda:
            raise _L.HipExtensionError("inputs['input'] is on %s; expected a ROCm device tensor" % inp.device)
        N, E, nt = inp.size(0), self.num_model, self.num_token
        T = self.max_face_length if parallel else self.num_labels
        mask = self.process_masks(inputs["input_mask"]).to(torch.bool)               # N x S, True = never a key / never selected
        val, pos, qpos = self.get_embeddings(inp.to(torch.float32), label)
        src, pos = self.patch_source(val, pos)                                          # S x N x E, S x 1 x E
        qpos = qpos.transpose(0, 1)[: T - 1]                                            # (T-1) x 1 x E
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos)                  # S x N x E
        mem_rows = memory.transpose(0, 1).contiguous()                                  # N x S x E: pointer keys, one copy per wireframe
        if parallel:
            ni_in = inputs["num_input"]
            num_input = [int(n) for n in (ni_in.tolist() if torch.is_tensor(ni_in) else ni_in)]
            F = max(num_input)
            start = torch.arange(F, device=inp.device, dtype=torch.long).repeat(N, 1)   # anchors WITHOUT the token offset
            for i, n in enumerate(num_input):
                start[i, n:] = nt - 1                                                    # padding anchors (model_para.py:204-205)
            tokens = start.reshape(1, N * F)
            mem_seq, mask_seq = memory.repeat_interleave(F, 1), mask.repeat_interleave(F, 0)
        else:
            F = 1
            tokens = torch.full((1, N), self.token.SOS, dtype=torch.long, device=inp.device)
            mem_seq, mask_seq = memory, mask
        mask_u8 = mask.to(torch.uint8).contiguous()
        extra = self._extra_mask(inputs)
        eos_seen, pointer = 0, None
        trace = getattr(self, "_module_trace", None)        # tests: a list that receives the masked logits of every step
        if trace is not None:
            self._module_memory = mem_rows
        for step in range(T - 1):
            tgt = torch.gather(mem_seq, 0, tokens.unsqueeze(-1).expand(-1, -1, E))        # (step+1) x B x E
            pointer = self.project(self.decoder(tgt, mem_seq, memory_key_padding_mask=mask_seq, pos=pos,
                                                query_pos=qpos[: step + 1]))
            res = ops.pointer_argmax(pointer[-1].contiguous(), mem_rows, mask=mask_u8, extra_mask=extra,
                                     seqs_per_group=F, want_logits=trace is not None)
            nxt = res["next"].to(torch.long).unsqueeze(0)
            if trace is not None:
                trace.append(res["logits"])
            tokens = torch.cat((tokens, nxt), dim=0)
            if parallel:
                if bool((nxt < nt).all()):
                    break
            else:
                eos_seen += int((nxt == self.token.EOS).sum())
                if eos_seen == N:
                    break
        pad = torch.zeros((T - tokens.size(0), tokens.size(1)), dtype=torch.long, device=inp.device)
        predict = torch.cat((tokens, pad), dim=0).transpose(0, 1)
        if parallel:
            inputs["predict"] = predict.reshape(N, F, T)
        else:
            inputs["embedding"] = mem_rows
            inputs["pointer"] = pointer.transpose(0, 1)
            inputs["predict"] = predict
        return inputs

    def _ln_in_epilogue(self):
        v = getattr(self, "x3_ln_in_epilogue", None)
        return (getattr(self, "split_kind", SPLIT_KIND_DEFAULT) == "bf16x3") if v is None else bool(v)

    def engine(self):
        """PathEngine bound to this module's parameters (rebuilt when they moved, e.g. after .to())."""
        self._check_supported()
        eng = self._engine_obj
        if eng is None or not eng.pointers_current() or eng.has_planes != (self.x3_min_rows > 0) or \
                eng.ln_in_epilogue != self._ln_in_epilogue() or \
                eng.requested_kind != getattr(self, "split_kind", SPLIT_KIND_DEFAULT):
            tensors = {k: v for k, v in self.state_dict(keep_vars=True).items() if v.dtype == torch.float32}
            dev = tensors["project.weight"].device
            if dev.type != "cuda":
                raise _L.HipExtensionError(
                    "model parameters are on %s: the faceformer_amd decode path runs only on a ROCm "
                    "device through libfaceformer_hip.so (no CPU fallback). Move the model with "
                    ".cuda()." % dev)
            eng = PathEngine(tensors, self.num_head, self.num_token, self.encoder.norm.eps,
                             bf16_split_planes=self.x3_min_rows > 0, fold_layernorm=True,
                             ln_in_epilogue=self._ln_in_epilogue(),
                             split_kind=getattr(self, "split_kind", SPLIT_KIND_DEFAULT))
            self._engine_obj = eng
        return eng

    def _encode(self, inputs, eng=None):
        """(engine, memory [N,S,E], mask_u8 [N,S], kv_len [N])"""
        inp, input_mask = inputs["input"], inputs["input_mask"]
        eng = self.engine() if eng is None else eng
        if not inp.is_cuda:
            raise _L.HipExtensionError("inputs['input'] is on %s; expected a ROCm device tensor" % inp.device)
        if input_mask.dtype in (torch.bool, torch.uint8) and input_mask.dim() == 2 and input_mask.is_cuda:
            mask, kv_len = eng.prepare_mask(input_mask)      # process_masks + key lengths in one launch
        else:
            mask, kv_len = self.process_masks(input_mask).to(torch.uint8).contiguous(), None
        memory, kv_len = eng.encode(inp.to(torch.float32).flatten(-2, -1), mask, kv_len)
        return eng, memory, mask, kv_len

    def _extra_mask(self, inputs):
        """Optional `inputs['extra_mask']` (bool, one row per decoded sequence, True = edge the pointer
        may not select, e.g. a co-edge adjacency rule; not a reference key): the never-masked
        special-token columns are prepended and the mask is OR-ed into the pointer's padding mask
        inside the kernel."""
        extra = inputs.get("extra_mask")
        if extra is None:
            return None
        pad = torch.zeros((extra.size(0), self.num_token), dtype=torch.bool, device=extra.device)
        return torch.cat([pad, extra.to(torch.bool)], dim=1).to(torch.uint8).contiguous()

    def forward(self, inputs):
        if self.training:
            return self.forward_train(inputs)
        return self.forward_eval(inputs)


__all__ = ["SurfaceFormerBase", "min_value_of_dtype"]
