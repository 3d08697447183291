"""GPU probe: what the partial-tile exchange of the stream-K projection kernel costs one block (needs the `skstamp` variant build:
tools/build_variant.sh skstamp -DFF_EXP_SK_STAMP).  Block 100 stamps the shader clock at entry, around its hand-over (the partial
tile it contributes, at the start of its range), around its fix-up (the tile it owns, at the end) and at its end."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L  # noqa: E402


def main():
    lib = L.load()
    raw = C.CDLL(L.LIB_PATH)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, K, N in ((7424, 512, 512), (7424, 512, 1536), (5120, 512, 512), (9216, 1024, 512)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            rc = lib.ff_gemm_f32(a.data_ptr(), K, None, 0, w.data_ptr(), K, b.data_ptr(), out.data_ptr(), N, out.data_ptr(), N, M, N, K, 0, 6, st)
            assert rc == 0, lib.ff_last_error()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)()
        assert raw.ff_exp_read_sk_stamps(buf) == 0
        t = [int(x) for x in buf]
        tot = t[5] - t[0]
        print("M=%d K=%d N=%d (tile 6 = stream-K): block 100 ran %d cycles; hand-over %d (at +%d), fix-up %d (at +%d), whole-tile epilogue stamp at +%d"
              % (M, K, N, tot, t[2] - t[1], t[1] - t[0], t[4] - t[3], t[3] - t[0], t[6] - t[0]))


if __name__ == "__main__":
    main()
