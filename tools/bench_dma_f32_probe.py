"""GPU: the f32 LDS-DMA projection kernel (plain, residual, LayerNorm-consuming) and the 2 x fp16 kernel on large decode shapes --
used with probe builds of ff_gemm_x3.hip (FF_HIP_LIB).   python tools/bench_dma_f32_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

row = []
for M, K, N in [(9216, 512, 1536), (9216, 512, 512), (9216, 1024, 512), (9216, 512, 1024), (32768, 512, 1536), (32768, 1024, 512)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    resid = out if N == 512 else None
    p2 = ops.split_weight(w, "fp16x2")
    fl = 2.0 * M * N * K
    it = max(5, min(100, int(2e11 / fl)))
    f32 = fl / timeit(lambda: ops.linear(a, w, b, residual=resid, out=out, tile=11), it) / 1e12
    h = fl / timeit(lambda: ops.linear_x3(a, p2, b, residual=resid, out=out), it) / 1e12
    row.append("%dx%d->%d %.0f/%.0f" % (M, K, N, f32, h))
print(os.environ.get("FF_HIP_LIB", "in-tree").split("/")[-1], "(f32 / fp16x2 TF/s-eq)", " | ".join(row))
