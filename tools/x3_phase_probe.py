"""GPU, probe build only (FF_VARIANT_SRC=ff_gemm_x3 tools/build_variant.sh stamp -DX3_EXP_STAMP; FF_HIP_LIB=build_ub/lib_stamp.so):
where wave 0 of workgroup 0 of the split projection kernel spends its K-loop iterations (shader-clock cycles per iteration):
the MFMA block with everything placed in its gaps / the wait for DMA + fragment reads / segment branch + block barrier / rest.
    python tools/x3_phase_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L  # noqa: E402
from faceformer_amd.hip import ops  # noqa: E402

raw = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
print("%-28s %-8s | %8s %8s %8s %8s | %8s %10s %12s" % ("shape", "form", "mfma blk", "wait", "barrier", "rest", "iters", "cyc/iter", "kernel cyc"))
for M, K, N in [(9216, 512, 1536), (9216, 512, 512), (32768, 512, 1536), (1024, 512, 1536)]:
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    st = torch.randn(M, K // 32, 2, device="cuda").abs()
    for kind in ("fp16x2", "bf16x3"):
        p = ops.split_weight(w, kind)
        for form in ("plain", "ln"):
            for _ in range(3):
                out = ops.linear_x3(a, p, b) if form == "plain" else ops.linear_x3_ln(a, p, b, stats_in=st)
            torch.cuda.synchronize()
            assert raw.ff_exp_read_x3_stamps(buf) == 0
            d = list(buf)
            n = max(1, d[4])
            print("%-28s %-8s | %8.0f %8.0f %8.0f %8.0f | %8d %10.0f %12d" % ("%dx%d->%d %s" % (M, K, N, kind), form, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4],
                                                                                   (d[0] + d[1] + d[2] + d[3]) / n, d[5]))
