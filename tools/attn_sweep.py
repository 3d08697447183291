"""Cross-attention (config B shapes, one wireframe) over decode steps and key counts, every kernel variant.
Back-to-back launches timed with events (queue depth hides the host): us per launch.
    python tools/attn_sweep.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L, ops  # noqa: E402

lib = L.load()
F, H, E = 256, 8, 512


def run(t, S, algo, iters=30, F=F):
    ops.set_attention_algo(algo)
    q = torch.randn(t * F, E, device="cuda")
    kv = torch.randn(S, 2 * E, device="cuda")
    out = torch.empty(t * F, E, device="cuda")
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * E, out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = E, 2 * E, 2 * E, E
    d.num_groups, d.num_heads, d.nq, d.nk = 1, H, F * t, S
    d.q_group_stride, d.q_inner, d.q_outer_stride = F, F, F   # (one wireframe: B = F)
    d.k_group_stride, d.k_stride = S, 1
    d.scale = 0.125
    st = torch.cuda.current_stream().cuda_stream
    lib.ff_attention(C.byref(d), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        lib.ff_attention(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


if "--seq" in sys.argv:   # single-sequence decode (seq2seq variant): t queries; cross key sets 68 / 220, self S = t
    print("%4s %5s | %8s %8s %8s %8s" % ("t", "S", "lds(1)", "wave(2)", "resid(3)", "auto(0)"))
    for S in (68, 220, 0):
        for t in (16, 32, 64, 96, 128, 192, 258):
            k = S if S else t
            print("%4d %5d | %8.1f %8.1f %8.1f %8.1f" % (t, k, run(t, k, 1, F=1), run(t, k, 2, F=1), run(t, k, 3, F=1),
                                                       run(t, k, 0, F=1)))
    ops.set_attention_algo(0)
    sys.exit(0)
if "--long" in sys.argv:   # key sets of the 512 / 1024-edge wireframes (config E): F = 512 / 1024 sequences of ONE wireframe
    print("%4s %5s %5s | %8s %8s %8s   TF/s: lds, wave, automatic choice" % ("t", "F", "S", "lds(1)", "wave(2)", "auto(0)"))
    for Fw, S in ((512, 516), (1024, 1028), (300, 304)):
        for t in (1, 2, 4, 8, 16, 24, 37):
            t1, t4, t0 = run(t, S, 1, iters=10, F=Fw), run(t, S, 2, iters=10, F=Fw), run(t, S, 0, iters=10, F=Fw)
            fl = 4.0 * 512 * t * Fw * S / 1e6
            print("%4d %5d %5d | %8.1f %8.1f %8.1f   %.1f %.1f %.1f" % (t, Fw, S, t1, t4, t0, fl / t1, fl / t4, fl / t0))
    ops.set_attention_algo(0)
    sys.exit(0)
def run_self(t, B, algo, iters=20):
    """decoder self-attention: B sequences, position-major packed q|k|v rows (j * B + b), keys = queries = t"""
    ops.set_attention_algo(algo)
    qkv = torch.randn(t * B, 3 * E, device="cuda")
    out = torch.empty(t * B, E, device="cuda")
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = qkv.data_ptr(), qkv.data_ptr() + 4 * E, qkv.data_ptr() + 8 * E, out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = 3 * E, 3 * E, 3 * E, E
    d.num_groups, d.num_heads, d.nq, d.nk = B, H, t, t
    d.q_group_stride, d.q_inner, d.q_outer_stride = 1, 1, B
    d.k_group_stride, d.k_stride = 1, B
    d.scale = 0.125
    st = torch.cuda.current_stream().cuda_stream
    lib.ff_attention(C.byref(d), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        lib.ff_attention(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def run_enc(G, S, algo, iters=20):
    """encoder self-attention: G wireframes of S tokens, group-major rows"""
    ops.set_attention_algo(algo)
    qkv = torch.randn(G * S, 3 * E, device="cuda")
    out = torch.empty(G * S, E, device="cuda")
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = qkv.data_ptr(), qkv.data_ptr() + 4 * E, qkv.data_ptr() + 8 * E, out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = 3 * E, 3 * E, 3 * E, E
    d.num_groups, d.num_heads, d.nq, d.nk = G, H, S, S
    d.q_group_stride, d.q_inner, d.q_outer_stride = S, S, 0
    d.k_group_stride, d.k_stride = S, 1
    d.scale = 0.125
    st = torch.cuda.current_stream().cuda_stream
    lib.ff_attention(C.byref(d), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        lib.ff_attention(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


if "--self" in sys.argv:   # is the automatic choice the best kernel?  decoder self-attention and encoder shapes
    print("decoder self-attention   %4s %6s | %8s %8s %8s" % ("t", "B", "lds(1)", "wave(2)", "auto(0)"))
    for B in (256, 4096):
        for t in (1, 4, 8, 16, 24, 32, 33, 36, 37):
            print("                         %4d %6d | %8.1f %8.1f %8.1f" % (t, B, run_self(t, B, 1), run_self(t, B, 2), run_self(t, B, 0)))
    for B in (1,):
        for t in (16, 64, 128, 258):
            print("                         %4d %6d | %8.1f %8.1f %8.1f" % (t, B, run_self(t, B, 1), run_self(t, B, 2), run_self(t, B, 0)))
    print("encoder self-attention   %4s %6s | %8s %8s %8s %8s" % ("G", "S", "lds(1)", "wave(2)", "resid(3)", "auto(0)"))
    for G, S in ((1, 68), (1, 260), (16, 260), (128, 260), (1, 516), (1, 1028), (2, 1028), (8, 132), (15, 68)):
        print("                         %4d %6d | %8.1f %8.1f %8.1f %8.1f" % (G, S, run_enc(G, S, 1), run_enc(G, S, 2), run_enc(G, S, 3), run_enc(G, S, 0)))
    ops.set_attention_algo(0)
    sys.exit(0)
print("%4s %5s | %8s %8s %8s %8s" % ("t", "S", "lds(1)", "wave(2)", "resid(3)", "auto(0)"))
for S in (32, 64, 132, 260):
    for t in (1, 4, 8, 16, 24, 32, 36):
        print("%4d %5d | %8.1f %8.1f %8.1f %8.1f" % (t, S, run(t, S, 1), run(t, S, 2), run(t, S, 3), run(t, S, 0)))
ops.set_attention_algo(0)
