"""Edge / position embeddings of the decode path (surface of reference `faceformer/embedding.py`).

`VanillaEmedding` [sic, the reference's spelling is part of the import surface] = four learned
special-token rows followed by MLP(100 -> E -> E) of every edge's flattened 50x2 polyline
(reference embedding.py:7-38); the two GEMMs (bias+ReLU / bias epilogues) run on the f32 MFMA kernel
and the concatenation is one row-copy kernel.  `PositionEmbeddingLearned` is a learned table sliced
by the *length* of its argument (reference embedding.py:90-108) -- a pure lookup, no arithmetic.
`PositionalEncoding` and `CoordinateEmbedding` are unused by the path (reference embedding.py:41-87)
and kept only so the module surface stays importable.
"""
import math

import torch
import torch.nn as nn

from .hip import lib as _L
from .hip import ops

__all__ = ["VanillaEmedding", "CoordinateEmbedding", "PositionalEncoding", "PositionEmbeddingLearned"]


class VanillaEmedding(nn.Module):
    def __init__(self, input_dim, num_model, token):
        super().__init__()
        self.num_tokens = token.len
        self.embedding_token = nn.Embedding(self.num_tokens, num_model)
        self.embedding_value = nn.Sequential(
            nn.Linear(input_dim, num_model), nn.ReLU(), nn.Linear(num_model, num_model))

    def embed_points(self, lines):
        return lines.flatten(-2, -1)

    def forward(self, coord):
        """coord: N x L x P x D  ->  N x (num_tokens + L) x E"""
        n, num_lines = coord.size(0), coord.size(1)
        flat = self.embed_points(coord).reshape(n * num_lines, -1).contiguous()
        fc1, fc2 = self.embedding_value[0], self.embedding_value[2]
        hidden = ops.linear(flat, fc1.weight, fc1.bias, act=1)
        edge = ops.linear(hidden, fc2.weight, fc2.bias)
        e = edge.size(1)
        out = torch.empty((n, self.num_tokens + num_lines, e), device=coord.device, dtype=torch.float32)
        _L.check(_L.load().ff_assemble_embedding(
            self.embedding_token.weight.data_ptr(), self.num_tokens, edge.data_ptr(), e, n, num_lines, e,
            out.data_ptr(), torch.cuda.current_stream().cuda_stream), "ff_assemble_embedding")
        return out


class PositionEmbeddingLearned(nn.Module):
    """Absolute learned position table; forward(x) returns rows [0, x.size(1)) as 1 x len x E."""

    def __init__(self, num_model, max_len=5000):
        super().__init__()
        self.register_buffer("position", torch.arange(0, max_len, dtype=torch.long).unsqueeze(0))
        self.pos_embed = nn.Embedding(max_len, num_model)
        nn.init.kaiming_normal_(self.pos_embed.weight, mode="fan_in")

    def forward(self, x):
        length = x.size(1)
        if length > self.pos_embed.num_embeddings:
            raise IndexError("sequence of length %d exceeds the position table (%d rows)"
                             % (length, self.pos_embed.num_embeddings))
        return self.pos_embed.weight[:length].unsqueeze(0)


class PositionalEncoding(nn.Module):
    """Fixed sinusoidal table (unused by the path)."""

    def __init__(self, num_model, max_len=5000):
        super().__init__()
        angle = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1) * torch.exp(
            torch.arange(0, num_model, 2).float() * (-math.log(10000.0) / num_model))
        pe = torch.zeros(max_len, num_model)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(angle), torch.cos(angle)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return self.pe[:, : x.size(1)]


class CoordinateEmbedding(nn.Module):
    """Quantised-coordinate embedding (unused by the path); lookup + one bias-free projection."""

    def __init__(self, num_axes, num_bits, num_embed, num_model, dependent_embed=False):
        super().__init__()
        ntoken = 2 ** num_bits if dependent_embed else 2 ** num_bits * num_axes
        self.embedding_token = nn.Embedding(3, num_model)
        self.embedding_value = nn.Embedding(ntoken, num_embed)
        self.linear_proj = nn.Linear(num_axes * num_embed, num_model, bias=False)

    def forward(self, coord):
        n, s, _ = coord.shape
        gathered = self.embedding_value.weight[coord].reshape(n * s, -1).contiguous()
        proj = ops.linear(gathered, self.linear_proj.weight).view(n, s, -1)
        tokens = self.embedding_token.weight.unsqueeze(0).expand(n, -1, -1)
        return torch.cat((tokens, proj), dim=1)
