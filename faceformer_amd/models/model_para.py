"""SurfaceFormer_Parallel: one token sequence per anchor edge, all decoded together (surface of
reference `faceformer/models/model_para.py`; greedy eval path 181-241, pointer head 173-179).

`forward_eval` runs on the native engine (ff_encode + ff_decode, variant FF_PARALLEL).  What the engine
keeps from the reference, quirks included: anchors are arange(F) WITHOUT the special-token offset and
padding anchors start from token num_token-1 (model_para.py:201-205); the decoder is re-run,
unmasked, over the whole prefix each step (222-223); masked logits are finfo.min and ties go to the
lowest index (173-179); the loop stops after the first step whose tokens are all special
(232-233) and the rest is zero padded (236).  What it does differently (results unchanged): memory
and masks are never replicated per sequence (212-214), cross-attention K/V are projected once, the
last layer and the output projection are evaluated for the newest position only.
"""
import torch

from ..hip import lib as _L
from .common import SurfaceFormerBase


class SurfaceFormer_Parallel(SurfaceFormerBase):

    def __init__(self, num_model=512, num_head=8, num_feedforward=2048, num_encoder_layers=6,
                 num_decoder_layers=6, dropout=0.1, activation="relu", normalize_before=True,
                 num_points_per_line=50, num_lines=64, point_dim=2, max_face_length=10, token=None,
                 teacher_forcing_ratio=0, **kwargs):
        super().__init__()
        self.max_face_length = max_face_length
        self._build(num_model, num_head, num_feedforward, num_encoder_layers, num_decoder_layers,
                    dropout, activation, normalize_before, num_points_per_line, num_lines, point_dim,
                    max_face_length, token, teacher_forcing_ratio)

    def get_embeddings(self, input, label):
        val_embed = self.val_enc(input)
        return val_embed, self.pos_enc(val_embed), self.query_pos_enc(label.transpose(1, 2))

    def forward_eval(self, inputs):
        """inputs: input N x L x P x D, input_mask N x L (True = padding), label N x F' x T (shape
        only), num_input: N edge counts.  Adds predict N x F x T (int64), F = max(num_input)."""
        label = inputs["label"]
        T = self.max_face_length
        if not self.engine_supported():      # post-norm / gelu constructor arguments: the sub-module loop (models/common.py)
            return self._forward_eval_modules(inputs, parallel=True)
        if label.size(2) < T - 1:
            raise ValueError("label has %d positions but max_face_length-1=%d query positions are "
                             "needed" % (label.size(2), T - 1))
        ni_in = inputs["num_input"]
        # (a device tensor of counts comes over in ONE copy: int() per element is a synchronising copy per wireframe)
        num_input = [int(n) for n in (ni_in.tolist() if torch.is_tensor(ni_in) else ni_in)]
        F = max(num_input)
        N = len(num_input)
        if N != inputs["input"].size(0):
            raise ValueError("num_input has %d entries for a batch of %d" % (N, inputs["input"].size(0)))
        extra = self._extra_mask(inputs)
        # Ragged batch: decode the wireframes sorted by edge count so that a micro-batch holds wireframes of
        # (nearly) the same width; wireframes are independent, the result rows are put back in batch order.
        order = None
        if self.sort_by_edges and extra is None and len(set(num_input)) > 1:
            order = sorted(range(N), key=lambda i: -num_input[i])
            idx = torch.tensor(order, device=inputs["input"].device)
            sub = {"input": inputs["input"].index_select(0, idx), "input_mask": inputs["input_mask"].index_select(0, idx)}
            ni = [num_input[i] for i in order]
            eng = self.engine()
            staged = eng.stage_num_input(ni)       # (before the encoder is enqueued: see stage_num_input)
            eng, memory, mask, kv_len = self._encode(sub, eng)
        else:
            ni = num_input
            eng = self.engine()
            staged = eng.stage_num_input(ni)
            eng, memory, mask, kv_len = self._encode(inputs, eng)
        out = eng.decode(memory, mask, kv_len, _L.FF_PARALLEL, T=T, F=F, num_input=ni, staged_num_input=staged,
                         chunk_wireframes=self.chunk_wireframes, chunk_seqs=self.chunk_seqs,
                         chunk_max_seqs=self.chunk_max_seqs,
                         num_streams=self.num_streams, sync_every=self.sync_every,
                         flags=self.decode_flags, x3_min_rows=self.x3_min_rows, ln_fuse_max_rows=self.ln_fuse_max_rows, extra_mask=extra)
        pred = out["predict"].view(N, F, T)
        if order is not None:
            inv = torch.empty(N, dtype=torch.long, device=pred.device)
            inv[torch.tensor(order, device=pred.device)] = torch.arange(N, device=pred.device)
            pred = pred.index_select(0, inv)
        inputs["predict"] = pred
        self.last_decode_stats = {"decoded_seqs": sum(min(F, n + 1) for n in num_input), "rows": N * F}
        return inputs
