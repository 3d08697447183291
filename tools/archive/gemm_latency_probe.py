"""GPU probe: is the f32 projection kernel waiting for its operands?  Times ff_gemm_f32 (stream-K / persistent, tile 7) on the
path's 9216-row shapes with the operands as they are and with every A row (or every W row) aliased onto ONE row (leading
dimension 0: the loads always hit the nearest cache).  Needs a library built with -DFF_EXP_NO_LD_CHECK (tools/build_variant.sh);
results are wrong by construction, timing only."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402


def main():
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    print("%7s %5s %5s | %9s %9s %9s %9s  (us per launch)" % ("M", "K", "N", "as is", "lda = 0", "ldw = 0", "both 0"))
    for M in (4096, 9216):
        for K, N in ((512, 1536), (512, 512), (512, 1024), (1024, 512)):
            a = torch.randn(M, K, device="cuda")
            w = torch.randn(N, K, device="cuda") * 0.05
            b = torch.randn(N, device="cuda")
            out = torch.empty(M, N, device="cuda")
            res = out if N == 512 else None
            act = 1 if N == 1024 else 0

            def run(lda, ldw):
                rc = lib.ff_gemm_f32(a.data_ptr(), lda, None, 0, w.data_ptr(), ldw, b.data_ptr(),
                                     res.data_ptr() if res is not None else None, N, out.data_ptr(), N, M, N, K, act, 7, st)
                assert rc == 0, lib.ff_last_error()
            iters = max(5, min(100, int(2e11 / (2.0 * M * N * K))))
            ts = [timeit(lambda: run(lda, ldw), iters) * 1e6 for lda, ldw in ((K, K), (0, K), (K, 0), (0, 0))]
            print("%7d %5d %5d | %9.1f %9.1f %9.1f %9.1f" % ((M, K, N) + tuple(ts)))


if __name__ == "__main__":
    main()
