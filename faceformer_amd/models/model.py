"""SurfaceFormer: all faces of a wireframe as ONE token sequence (surface of reference
`faceformer/models/model.py`; greedy eval path 169-219, pointer head 161-167).

Constructor arguments, `state_dict` layout and the `forward(inputs: dict) -> dict` contract match the
reference; `forward_eval` runs on the native engine (ff_encode + ff_decode, variant FF_SEQ2SEQ):
start token SOS, at most label_seq_length-1 steps, stop when the cumulative EOS count equals the
batch size (reference model.py:191,207-210), zero padding afterwards.
"""
import torch

from ..hip import lib as _L
from .common import SurfaceFormerBase


class SurfaceFormer(SurfaceFormerBase):

    def __init__(self, num_model=512, num_head=8, num_feedforward=2048, num_encoder_layers=6,
                 num_decoder_layers=6, dropout=0.1, activation="relu", normalize_before=True,
                 num_points_per_line=50, num_lines=1000, point_dim=2, label_seq_length=2000,
                 token=None, teacher_forcing_ratio=0, **kwargs):
        super().__init__()
        self.num_labels = label_seq_length
        self._build(num_model, num_head, num_feedforward, num_encoder_layers, num_decoder_layers,
                    dropout, activation, normalize_before, num_points_per_line, num_lines, point_dim,
                    label_seq_length, token, teacher_forcing_ratio)
        # One sequence per wireframe: a micro-batch is cut by sequences, not by `chunk_wireframes` wireframes.  256 sequences
        # x (label_seq_length - 1) positions = 66 k active rows at the last step of configs A / D (1.6 GB of scratch).
        self.chunk_max_seqs = 256
        # False: the reference's batch rule -- stop when the CUMULATIVE count of EOS tokens equals the batch size
        # (model.py:191,207-210; a sample that repeats its EOS can end a batch before another sample has produced its own).
        # True: stop at the first step by which EVERY wireframe has produced an EOS (FF_STOP_EACH_EOS) -- what a caller needs
        # when the records of a batch must equal those of one-wireframe decodes (main.py --batch-size, dist.decode_to_face_json).
        self.stop_each_eos = False

    def get_embeddings(self, input, label):
        val_embed = self.val_enc(input)
        return val_embed, self.pos_enc(val_embed), self.query_pos_enc(label)

    def forward_eval(self, inputs):
        """inputs: input N x L x P x D, input_mask N x L (True = padding), label N x T (shape only).
        Adds predict N x T (int64), embedding N x S x E, pointer N x t_last x E."""
        label = inputs["label"]
        T = self.num_labels
        if inputs["input"].size(0) == 0:     # an empty batch: the reference's loop stops after its first step (0 EOS == batch size 0, model.py:207)
            dev, S = inputs["input"].device, inputs["input"].size(1) + self.num_token
            inputs["embedding"] = torch.zeros((0, S, self.num_model), device=dev)
            inputs["pointer"] = torch.zeros((0, 1, self.num_model), device=dev)
            inputs["predict"] = torch.zeros((0, T), dtype=torch.long, device=dev)
            return inputs
        if not self.engine_supported():      # post-norm / gelu constructor arguments: the sub-module loop (models/common.py)
            return self._forward_eval_modules(inputs, parallel=False)
        if label.size(1) < T - 1:
            raise ValueError("label has %d positions but label_seq_length-1=%d query positions are "
                             "needed" % (label.size(1), T - 1))
        eng, memory, mask, kv_len = self._encode(inputs)
        out = eng.decode(memory, mask, kv_len, _L.FF_SEQ2SEQ, T=T, F=1,
                         chunk_wireframes=self.chunk_wireframes, chunk_seqs=self.chunk_seqs,
                         chunk_max_seqs=self.chunk_max_seqs, num_streams=self.num_streams, sync_every=1,
                         flags=self.decode_flags | (_L.FF_STOP_EACH_EOS if self.stop_each_eos else 0),
                         x3_min_rows=self.x3_min_rows, ln_fuse_max_rows=self.ln_fuse_max_rows, tok_sos=self.token.SOS, tok_eos=self.token.EOS,
                         return_pointer=True, extra_mask=self._extra_mask(inputs))
        inputs["embedding"] = memory
        inputs["pointer"] = out["pointer"].transpose(0, 1)
        inputs["predict"] = out["predict"]
        return inputs
