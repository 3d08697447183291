"""GPU micro-benchmark: the LayerNorm-fused GEMM forms against the plain kernel on the decode path's shapes.
    plain   : ff_gemm_f32 (residual for N=512)
    stats   : ff_gemm_f32_ln emitting row statistics (producer; N=512 shapes)
    ln-in   : ff_gemm_f32_ln normalising A from statistics (+ position table for N=512/1536) (consumer; K=512)
    ln+gemm : standalone ff_layernorm followed by the plain GEMM (what the fusion replaces)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402


def main():
    dev = "cuda"
    ts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16,24,36").split(",")]
    print("%7s %5s %5s | %8s %8s %8s %8s   (us per launch)" % ("M", "K", "N", "plain", "stats", "ln-in", "ln+gemm"))
    for t in ts:
        M = 256 * t
        for K, N in [(512, 1536), (512, 512), (512, 1024), (1024, 512)]:
            a = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev)
            gam, bet = torch.randn(K, device=dev), torch.randn(K, device=dev)
            pos = torch.randn(64, K, device=dev)
            out = torch.empty(M, N, device=dev)
            resid = out if N == 512 else None
            act = 1 if N == 1024 else 0
            flops = 2.0 * M * N * K
            iters = max(5, min(100, int(2e11 / flops)))
            tp = timeit(lambda: ops.linear(a, w, b, act=act, residual=resid, out=out), iters)
            ts_ = tl = tg = float("nan")
            if N == 512:
                ts_ = timeit(lambda: ops.linear_ln(a, w, b, residual=resid, want_stats=True, out=out), iters)
            if K == 512:
                st = torch.randn(M, K // 32, 2, device=dev).abs()
                tab = torch.randn(64, N, device=dev) if N != 1024 else None
                tl = timeit(lambda: ops.linear_ln(a, w, b, act=act, stats_in=st, row_table=tab, row_div=256,
                                                  row_cols=min(N, 1024), out=out), iters)
                tg = timeit(lambda: (ops.layernorm(a, gam, bet, pos=pos, pos_div=256, pos_mod=64),
                                     ops.linear(a, w, b, act=act, out=out)), iters)
            print("%7d %5d %5d | %8.1f %8.1f %8.1f %8.1f" % (M, K, N, tp * 1e6, ts_ * 1e6, tl * 1e6, tg * 1e6))


if __name__ == "__main__":
    main()
