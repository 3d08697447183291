"""GPU micro-benchmark: the LayerNorm-folded forms of the 3 x bf16 kernel (MODE 1 consumer, MODE 2 producer) against the plain
form on the decode path's projections.  TF/s fp32-equivalent.    python tools/bench_gemm_x3_ln.py [--ms 4096,9216,32768]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="3072,6144,9216,32768,65536")
    args = ap.parse_args()
    E, FF = 512, 1024
    print("%8s | %-22s | %8s %8s %8s | %8s %8s %8s" % ("M", "projection", "plain", "folded", "fold-epi", "f32 fold", "f32 dma", "dma fold"))
    for M in [int(v) for v in args.ms.split(",")]:
        x = torch.randn(M, E, device="cuda")
        h = torch.randn(M, FF, device="cuda")
        stats = torch.stack([x.view(M, 16, 32).mean(2), ((x.view(M, 16, 32) - x.view(M, 16, 32).mean(2, keepdim=True)) ** 2).sum(2)], 2).contiguous()
        tab = torch.randn((M + 255) // 256, 1024, device="cuda")
        for name, N, K, mode in (("q|k|v (LN in, table)", 1536, E, 1), ("cross-q (LN in, table)", E, E, 1), ("linear1 (LN in, relu)", FF, E, 1),
                                 ("out-proj (stats out)", E, E, 2), ("linear2 (stats out)", E, FF, 2)):
            w = torch.randn(N, K, device="cuda") * 0.05
            b = torch.randn(N, device="cuda")
            planes = ops.split_weight(w)
            a = x if K == E else h
            out = torch.empty(M, N, device="cuda")
            res = torch.randn(M, N, device="cuda") if mode == 2 else None
            flops = 2.0 * M * N * K
            iters = max(3, min(50, int(2e11 / flops)))
            act = 1 if N == FF else 0
            tp = timeit(lambda: ops.linear_x3(a, planes, b, act=act, residual=res, out=out), iters)
            if mode == 1:
                t_ = tab[:, :min(N, 1024)].contiguous() if N != FF else None
                tf = timeit(lambda: ops.linear_x3_ln(a, planes, b, act=act, stats_in=stats, row_table=t_, row_div=256,
                                                     row_cols=(t_.size(1) if t_ is not None else 0), out=out), iters)
                t32 = timeit(lambda: ops.linear_ln(a, w, b, act=act, stats_in=stats, row_table=t_, row_div=256,
                                                   row_cols=(t_.size(1) if t_ is not None else 0), out=out), iters)
                t11 = timeit(lambda: ops.linear_ln(a, w, b, act=act, stats_in=stats, row_table=t_, row_div=256,
                                                   row_cols=(t_.size(1) if t_ is not None else 0), out=out, tile=11), iters)
                cs = w.double().sum(1).float().contiguous()
                te = timeit(lambda: ops.linear_x3_ln(a, planes, b, act=act, stats_in=stats, row_table=t_, row_div=256,
                                                     row_cols=(t_.size(1) if t_ is not None else 0), out=out, colsum=cs), iters)
            else:
                tf = timeit(lambda: ops.linear_x3_ln(a, planes, b, residual=res, want_stats=True, out=out), iters)
                t32 = timeit(lambda: ops.linear_ln(a, w, b, residual=res, want_stats=True, out=out), iters)
                t11 = timeit(lambda: ops.linear_ln(a, w, b, residual=res, want_stats=True, out=out, tile=11), iters)
                te = tf
            tpl = timeit(lambda: ops.linear(a, w, b, act=act, residual=res, out=out, tile=11), iters)
            print("%8d | %-22s | %8.1f %8.1f %8.1f | %8.1f %8.1f %8.1f" % (M, name, flops / tp / 1e12, flops / tf / 1e12, flops / te / 1e12,
                                                                          flops / t32 / 1e12, flops / tpl / 1e12, flops / t11 / 1e12))


if __name__ == "__main__":
    main()
