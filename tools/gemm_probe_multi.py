"""Launch each (M,N,K,tile) config a few times under rocprofv3 --kernel-trace: GPU-side durations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402

tiles = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
Ms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256,1024,4096,9216").split(",")]
cfgs = []
for M in Ms:
    for (K, N) in ((512, 512), (512, 1536), (1024, 512)):
        for tile in tiles:
            cfgs.append((M, N, K, tile))
for (M, N, K, tile) in cfgs:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(6):
        ops.linear(a, w, b, tile=tile, out=out, residual=out if N == 512 else None)
    torch.cuda.synchronize()
print("CFGS", cfgs)
