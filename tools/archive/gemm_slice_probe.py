"""GPU probe: where a K-slice of the persistent f32 projection kernel spends its time.  Needs the `stamp` variant build
(tools/build_variant.sh stamp -DFF_EXP_STAMP): the four waves of workgroup 0 record the shader clock in front of and behind the
block barrier of their first 256 slices.  Prints, per wave, the cycles per slice, the share spent waiting in the barrier and the
MFMA-issue floor (16 MFMAs x 64 cycles per wave, two waves per SIMD)."""
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
    cases = [(9216, 512, 1536, 0), (9216, 512, 512, 0), (9216, 1024, 512, 0), (32768, 512, 1536, 0),
             (9216, 512, 1536, 1), (9216, 512, 512, 2)]   # mode 1: LayerNorm-normalising consumer, 2: statistics-emitting producer
    for M, K, N, mode in cases:
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        stats = torch.randn(M, max(K, N) // 32, 2, device="cuda").abs()
        for _ in range(3):
            if mode == 0:
                rc = lib.ff_gemm_f32(a.data_ptr(), K, None, 0, w.data_ptr(), K, b.data_ptr(), None, 0, out.data_ptr(), N, M, N, K, 0, 3, st)
            else:
                d = L.GemmLnDesc()
                d.A, d.lda, d.W, d.ldw, d.bias = a.data_ptr(), K, w.data_ptr(), K, b.data_ptr()
                d.C, d.ldc, d.M, d.N, d.K, d.act, d.tile = out.data_ptr(), N, M, N, K, 0, 3
                if mode == 1:
                    d.ln_stats_in, d.ln_nseg, d.ln_eps = stats.data_ptr(), K // 32, 1e-5
                else:
                    d.residual, d.ldr, d.ln_stats_out = out.data_ptr(), N, stats.data_ptr()
                rc = lib.ff_gemm_f32_ln(C.byref(d), st)
            assert rc == 0, lib.ff_last_error()
        torch.cuda.synchronize()
        print("mode %d:" % mode, end=" ")
        buf = (C.c_ulonglong * (4 * 256 * 2))()
        assert raw.ff_exp_read_stamps(buf) == 0
        t = np.frombuffer(buf, dtype=np.uint64).reshape(4, 256, 2).astype(np.int64)
        nsl = K // 32
        tiles = (M // 64) * (N // 64)
        slices = min(256, (tiles + 511) // 512 * nsl)   # slices workgroup 0 ran (512 workgroups)
        print("M=%d K=%d N=%d: workgroup 0 ran %d slices (%d recorded)" % (M, K, N, (tiles + 511) // 512 * nsl, slices))
        for wv in range(4):
            pre, post = t[wv, :slices, 0], t[wv, :slices, 1]
            per = np.diff(post)                         # barrier exit -> next barrier exit
            wait = (post - pre)[1:]
            body = per - wait
            # steady state: skip the first tile's first slices
            s0 = 4
            if wv == 0 and slices >= 2 * nsl:   # the slices of the second tile, one by one
                print("     second tile, slice by slice:", " ".join("%d" % x for x in per[nsl - 1: 2 * nsl - 1]))
            print("  wave %d: %7.0f cycles per slice (median; mean %.0f), barrier wait %5.0f (%.0f %%), body %6.0f; epilogue slices (every %d): %.0f"
                  % (wv, np.median(per[s0:]), per[s0:].mean(), np.median(wait[s0:]), 100 * wait[s0:].sum() / per[s0:].sum(),
                     np.median(body[s0:]), nsl, np.median(per[nsl - 1::nsl]) if len(per) >= nsl else float("nan")))
    print("MFMA floor: 16 x 64 = 1024 cycles per slice and wave alone on its SIMD, 2048 with two waves per SIMD (s_memtime counts"
          " at a constant 100 MHz-derived rate on some parts: compare the ratios, not the absolute numbers)")


if __name__ == "__main__":
    main()
