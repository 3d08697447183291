"""Run a few ff_gemm_f32 launches of one shape/tile (for rocprofv3 --pmc passes).
    python tools/gemm_probe.py M N K tile [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402

M, N, K, tile = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
a = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda")
for _ in range(iters):
    ops.linear(a, w, b, tile=tile, out=out)
torch.cuda.synchronize()
