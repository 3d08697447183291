"""Decoder self-attention (256 sequences x 8 heads, t positions) with the K / V rows of a sequence laid out position-major
(row = j * B + b: what the decode has, 1.5 MB between two keys of a sequence) against sequence-major (row = b * T + j:
contiguous keys).  Same kernel, same arithmetic; only the descriptor's strides differ.  us per launch.
    python tools/attn_layout_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402


def timeit(fn, iters=200):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


B, H, E, T = 256, 8, 512, 37
print("%4s | %14s %14s %14s" % ("t", "position-major", "kv seq-major", "qkv seq-major"))
for t in (1, 4, 8, 12, 16, 20, 24, 28, 32, 33, 36):
    qkv = torch.randn(t * B, 3 * E, device="cuda")          # position-major [t, B, 3E]
    out = torch.empty(t * B, E, device="cuda")
    kvs = torch.randn(B * T, 2 * E, device="cuda")          # sequence-major K | V [B, T, 2E]
    qs = torch.randn(B * T, E, device="cuda")               # sequence-major q
    outs = torch.empty(B * T, E, device="cuda")
    # flush-ish: touch a large buffer between timings is not done -- the decode's producer GEMM leaves the rows in L2 / MALL too
    pm = timeit(lambda: ops.attention(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], num_groups=B, num_heads=H, nq=t, nk=t,
                                      q_group_stride=1, q_inner=1, q_outer_stride=B, k_group_stride=1, k_stride=B, out=out))
    km = timeit(lambda: ops.attention(qkv[:, :E], kvs[:, :E], kvs[:, E:], num_groups=B, num_heads=H, nq=t, nk=t,
                                      q_group_stride=1, q_inner=1, q_outer_stride=B, k_group_stride=T, k_stride=1, out=out))
    sm = timeit(lambda: ops.attention(qs, kvs[:, :E], kvs[:, E:], num_groups=B, num_heads=H, nq=t, nk=t,
                                      q_group_stride=T, q_inner=t, q_outer_stride=0, k_group_stride=T, k_stride=1, out=outs))
    print("%4d | %14.1f %14.1f %14.1f" % (t, pm, km, sm))
