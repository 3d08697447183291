"""GPU probe: the phases of ONE small-M projection launch (gemm_panel_kernel, 256 rows x 512 -> 512: one 32x32 tile per CU).  Needs the
`pstamp` variant build (tools/build_variant.sh pstamp -DFF_EXP_PANEL_STAMP).  Workgroup 0 stamps the shader clock at entry, panels in
LDS, MFMA chains done, before / after its stores; the launch-to-launch period of a dependent chain of such launches is measured
with events beside it -- the difference is what the launch costs outside the kernel's own instructions."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402


def main():
    lib = L.load()
    raw = C.CDLL(L.LIB_PATH)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for M, K, N in ((256, 512, 512), (64, 512, 512), (256, 512, 1536), (256, 1024, 512)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        b = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")

        def run():
            rc = lib.ff_gemm_f32(a.data_ptr(), K, None, 0, w.data_ptr(), K, b.data_ptr(), None, 0, out.data_ptr(), N, M, N, K, 0, 7, st)
            assert rc == 0, lib.ff_last_error()
        period = timeit(run, 200) * 1e6
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 8)()
        assert raw.ff_exp_read_panel_stamps(buf) == 0
        t = np.array(list(buf)[:5], dtype=np.int64)
        d = np.diff(t)
        print("M=%d K=%d N=%d: period of back-to-back launches %.2f us; inside workgroup 0 [cycles]: entry -> panels in LDS %d, "
              "-> MFMA done %d, -> values ready %d, -> stores issued %d; total %d cycles = %.2f us at 2.4 GHz"
              % (M, K, N, period, d[0], d[1], d[2], d[3], t[4] - t[0], (t[4] - t[0]) / 2400.0))


if __name__ == "__main__":
    main()
