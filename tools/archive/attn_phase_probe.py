"""GPU probe: the phases of ONE K/V-resident cross-attention launch of config B (256 sequences, 260 keys) at decode step t.  Needs
the `astamp` variant build (tools/build_variant.sh astamp -DFF_EXP_ATTN_STAMP; build_variant.sh compiles ff_gemm.hip only, so the
script below rebuilds ff_attention.o into the variant itself)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L, ops  # noqa: E402

lib = L.load()
raw = C.CDLL(L.LIB_PATH)
F, H, E, S = 256, 8, 512, 260
print("%4s | %9s | cycles in workgroup 0, wave 0: %10s %10s %10s %10s %10s | total us @2.4GHz" % ("t", "launch us", "K/V in LDS", "items", "barrier", "merge+store", "barrier"))
for t in (4, 8, 16, 24, 32, 36):
    ops.set_attention_algo(3)
    q = torch.randn(t * F, E, device="cuda")
    kv = torch.randn(S, 2 * E, device="cuda")
    out = torch.empty(t * F, E, device="cuda")
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o = q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * E, out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = E, 2 * E, 2 * E, E
    d.num_groups, d.num_heads, d.nq, d.nk = 1, H, F * t, S
    d.q_group_stride, d.q_inner, d.q_outer_stride = F, F, F
    d.k_group_stride, d.k_stride = S, 1
    d.scale = 0.125
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.ff_attention(C.byref(d), st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        lib.ff_attention(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 24)()
    assert raw.ff_exp_read_attn_stamps(buf) == 0
    ts = np.array(list(buf)[:6], dtype=np.int64)
    dd = np.diff(ts)
    we = np.array(list(buf)[8:16], dtype=np.int64) - ts[1]
    print("      items done per wave [cycles after K/V landed]:", " ".join("%d" % x for x in we), " items per wave:", list(buf)[16:24])
    print("%4d | %9.1f | %41d %10d %10d %10d %10d | %.1f" % (t, a.elapsed_time(b) / 30 * 1e3, dd[0], dd[1], dd[2], dd[3], dd[4], (ts[5] - ts[0]) / 2400.0))
ops.set_attention_algo(0)
