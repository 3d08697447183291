"""GPU: 512- / 1024-column projections of the middle decode steps (rows = 256 t) on the 64x64 family (automatic choice below the
LDS-DMA thresholds), the LDS-DMA kernel with 64 x 128 tiles (tile 11) and with 64 x 64 tiles (tile 12): plain + residual, the
statistics-producing and the LayerNorm-consuming form.   python tools/bench_dma64.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

print("%6s %5s %5s | %-22s | %-22s | %-22s   (us per launch: family-7 / tile 11 / tile 12)" % ("M", "K", "N", "plain+res", "stats out", "ln in"))
ops.set_tuning("FF_DMA_MIN_ROWS", 1 << 30); ops.set_tuning("FF_DMA_MIN_ROWS_N512", 1 << 30); ops.set_tuning("FF_DMA_MIN_ROWS_WIDE", 1 << 30)
for t in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,8,9,12,16,20,24,25,28,31,33,36").split(",")]:
    M = 256 * t
    for K, N in [(512, 512), (1024, 512), (512, 1024), (512, 1536)]:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        res = torch.randn(M, N, device="cuda") if N == 512 else None
        st = torch.randn(M, K // 32, 2, device="cuda").abs() if K == 512 else None
        it = max(10, min(200, int(4e10 / (2.0 * M * N * K))))
        cols = []
        for form in ("plain", "stats", "ln"):
            r = []
            for tile in (0, 11, 12):
                if form == "plain":
                    f = lambda: ops.linear(a, w, b, residual=res, act=1 if N == 1024 else 0, out=out, tile=tile)
                elif form == "stats":
                    if N != 512:
                        r.append(float("nan")); continue
                    f = lambda: ops.linear_ln(a, w, b, residual=res, want_stats=True, tile=tile, out=out)
                else:
                    if st is None:
                        r.append(float("nan")); continue
                    f = lambda: ops.linear_ln(a, w, b, act=1 if N == 1024 else 0, stats_in=st, tile=tile, out=out)
                r.append(timeit(f, it) * 1e6)
            cols.append("%6.1f %6.1f %6.1f" % tuple(r))
        print("%6d %5d %5d | %s | %s | %s" % (M, K, N, cols[0], cols[1], cols[2]))
