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
        self.x3_ln_in_epilogue = True
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
    def _check_supported(self):
        if not self.normalize_before:
            raise NotImplementedError("the native decode engine implements the pre-norm path "
                                      "(normalize_before=True, what every reference config uses)")
        if self.activation_name != "relu":
            raise NotImplementedError("the native decode engine implements relu feed-forward layers")

    def engine(self):
        """PathEngine bound to this module's parameters (rebuilt when they moved, e.g. after .to())."""
        self._check_supported()
        eng = self._engine_obj
        if eng is None or not eng.pointers_current() or eng.has_planes != (self.x3_min_rows > 0) or \
                eng.ln_in_epilogue != bool(getattr(self, "x3_ln_in_epilogue", True)):
            tensors = {k: v for k, v in self.state_dict(keep_vars=True).items() if v.dtype == torch.float32}
            dev = tensors["project.weight"].device
            if dev.type != "cuda":
                raise _L.HipExtensionError(
                    "model parameters are on %s: the faceformer_amd decode path runs only on a ROCm "
                    "device through libfaceformer_hip.so (no CPU fallback). Move the model with "
                    ".cuda()." % dev)
            eng = PathEngine(tensors, self.num_head, self.num_token, self.encoder.norm.eps,
                             bf16_split_planes=self.x3_min_rows > 0, fold_layernorm=True,
                             ln_in_epilogue=getattr(self, "x3_ln_in_epilogue", True))
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
