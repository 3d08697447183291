"""Cross-attention launch of decode step t (config B shapes) for timing / rocprofv3 --pmc passes.
    python tools/attn_probe.py [t=36] [algo=0] [iters=10]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L, ops  # noqa: E402

t = int(sys.argv[1]) if len(sys.argv) > 1 else 36
algo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ops.set_attention_algo(algo)
F, S, H, E = 256, 260, 8, 512
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
lib, st = L.load(), torch.cuda.current_stream().cuda_stream
lib.ff_attention(C.byref(d), st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    lib.ff_attention(C.byref(d), st)
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / iters * 1e3
print("t=%d algo=%d: %.1f us, %.1f TF/s nominal" % (t, algo, us, 4 * 64 * H * F * t * S / us / 1e6))
