"""GPU: the 2 x fp16 projection kernel alone (plain + LayerNorm-consuming form) on a few decode shapes -- used with probe builds of
ff_gemm_x3.hip (tools/build_variant.sh, FF_HIP_LIB) that drop one ingredient of the K loop (operand DMA, split arithmetic, MFMAs).
    python tools/bench_x2h_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

row = []
for M, K, N in [(9216, 512, 1536), (9216, 512, 512), (9216, 1024, 512), (32768, 512, 1536), (32768, 512, 512)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    p2 = ops.split_weight(w, "fp16x2")
    fl = 2.0 * M * N * K
    it = max(5, min(100, int(2e11 / fl)))
    r = fl / timeit(lambda: ops.linear_x3(a, p2, b, out=out), it) / 1e12
    ln = float("nan")
    if K == 512:
        st = torch.randn(M, K // 32, 2, device="cuda").abs()
        ln = fl / timeit(lambda: ops.linear_x3_ln(a, p2, b, stats_in=st, out=out), it) / 1e12
    row.append("%dx%d->%d %.0f/%.0f" % (M, K, N, r, ln))
print(os.environ.get("FF_HIP_LIB", "in-tree").split("/")[-1], " | ".join(row))
