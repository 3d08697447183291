"""GPU: TF/s-equivalent of the split projection kernel with three bf16 terms (six products) and two fp16 terms (three products),
beside the f32 LDS-DMA / 64x64 family, over the decode path's shapes (M = 256 t rows): plain form (residual / ReLU as on the path)
and the LayerNorm-consuming forms (normalise first, normalise in the epilogue).
    python tools/bench_split_kinds.py [ts]            -> profiles/r06/gemm_split_kinds.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402


def main():
    dev = "cuda"
    ts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,9,16,24,36,64,128").split(",")]
    print("%7s %5s %5s | %9s %9s %9s | %9s %9s %9s %9s   (TF/s-equivalent = 2MNK / time)" % (
        "M", "K", "N", "f32", "bf16x3", "fp16x2", "ln bf16x3", "ln fp16x2", "epi bf16", "epi fp16"))
    for t in ts:
        M = 256 * t
        for K, N in [(512, 1536), (512, 512), (512, 1024), (1024, 512)]:
            a = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev)
            out = torch.empty(M, N, device=dev)
            resid = out if N == 512 else None
            act = 1 if N == 1024 else 0
            fl = 2.0 * M * N * K
            iters = max(5, min(100, int(2e11 / fl)))
            p3, p2 = ops.split_weight(w, "bf16x3"), ops.split_weight(w, "fp16x2")
            r = [fl / timeit(lambda: ops.linear(a, w, b, act=act, residual=resid, out=out), iters) / 1e12,
                 fl / timeit(lambda: ops.linear_x3(a, p3, b, act=act, residual=resid, out=out), iters) / 1e12,
                 fl / timeit(lambda: ops.linear_x3(a, p2, b, act=act, residual=resid, out=out), iters) / 1e12]
            ln = [float("nan")] * 4
            if K == 512:
                st = torch.randn(M, K // 32, 2, device=dev).abs()
                cs = w.double().sum(dim=1).float().contiguous()
                for i, (pl, c) in enumerate(((p3, None), (p2, None), (p3, cs), (p2, cs))):
                    ln[i] = fl / timeit(lambda: ops.linear_x3_ln(a, pl, b, act=act, stats_in=st, colsum=c, out=out), iters) / 1e12
            print("%7d %5d %5d | %9.1f %9.1f %9.1f | %9.1f %9.1f %9.1f %9.1f" % ((M, K, N) + tuple(r) + tuple(ln)))


if __name__ == "__main__":
    main()
