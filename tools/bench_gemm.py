"""GPU micro-benchmark of ff_gemm_f32 over the decode path's shapes (M = 256*t rows), every tile
config, against torch.mm (hipBLASLt/rocBLAS) on the same data.  Prints TF/s; run on the GPU box:
    python tools/bench_gemm.py [--ts 1,2,4,8,16,24,36] [--batch 256]
"""
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
    ap.add_argument("--ts", default="1,2,4,8,12,16,24,36")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tiles", default="3,7,8")
    ap.add_argument("--x3-variants", action="store_true",
                    help="time both launch shapes of the 3 x bf16 kernel next to the cost model's choice")
    ap.add_argument("--data", default="randn", choices=["randn", "zeros", "ones"],
                    help="operand values (the chip clocks to its power budget: zeros show the issue-bound rate)")
    args = ap.parse_args()
    dev = "cuda"
    shapes = [(512, 1536), (512, 512), (512, 1024), (1024, 512)]
    tiles = [int(x) for x in args.tiles.split(",")]
    print("%8s %5s %5s | %s | %8s" % ("M", "K", "N", " ".join("tile%d TF/s" % t for t in tiles), "torch.mm"))
    for t in [int(x) for x in args.ts.split(",")]:
        M = t * args.batch
        for K, N in shapes:
            a = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev)
            if args.data != "randn":
                val = 0.0 if args.data == "zeros" else 1.0
                a.fill_(val); w.fill_(val); b.fill_(val)
            out = torch.empty(M, N, device=dev)
            flops = 2.0 * M * N * K
            iters = max(3, min(50, int(2e11 / flops)))
            res = []
            # as on the path: N=512 products accumulate into the residual stream (residual aliases
            # the output), N=1024 is the ReLU feed-forward layer, N=1536 the plain q|k|v projection
            resid = out if N == 512 else None
            act = 1 if N == 1024 else 0
            for tile in tiles:
                dt = timeit(lambda: ops.linear(a, w, b, act=act, residual=resid, tile=tile, out=out), iters)
                res.append(flops / dt / 1e12)
                out.normal_()
            planes = ops.split_weight(w)
            dt3 = timeit(lambda: ops.linear_x3(a, planes, b, act=act, residual=resid, out=out), iters)
            out.normal_()
            var = ""
            if args.x3_variants:
                vs = []
                for shp in (1, 2):
                    ops.set_x3_tuning(shp)
                    vs.append(flops / timeit(lambda: ops.linear_x3(a, planes, b, act=act, residual=resid, out=out), iters) / 1e12)
                    out.normal_()
                ops.set_x3_tuning(0)
                var = " [whole %5.1f ranges %5.1f]" % tuple(vs)
            dt = timeit(lambda: torch.addmm(b, a, w.t(), out=out), iters)
            print("%8d %5d %5d | %s | x3 %6.1f%s | %8.1f" % (M, K, N, " ".join("%10.1f" % r for r in res),
                                                           flops / dt3 / 1e12, var, flops / dt / 1e12))


if __name__ == "__main__":
    main()
