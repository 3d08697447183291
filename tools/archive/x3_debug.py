"""Debug helper (GPU box): ff_gemm_x3_ln producer form over tunings; prints where the output leaves the fp64 reference."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops

def run(M, N, K, want_stats, tune):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    res = (3.0 + 2.0 * torch.randn(M, N, generator=g)).cuda()
    ops.set_x3_tuning(*tune)
    planes = ops.split_weight(W)
    ref = A.double() @ W.double().t() + b.double() + res.double()
    for rep in range(3):
        if want_stats:
            out, stats = ops.linear_x3_ln(A, planes, b, residual=res, want_stats=True)
        else:
            out = ops.linear_x3(A, planes, b, residual=res)
        torch.cuda.synchronize()
        bad = ((out.double() - ref).abs() > 1e-3 * ref.abs().max())
        nb = int(bad.sum())
        msg = ""
        if nb:
            idx = bad.nonzero()
            rows = idx[:, 0].unique()
            cols = idx[:, 1].unique()
            msg = " rows %s.. (%d) cols %s.. (%d) tiles128 m %s n %s" % (rows[:4].tolist(), len(rows), cols[:4].tolist(), len(cols),
                                                                      (rows // 128).unique()[:8].tolist(), (cols // 128).unique().tolist())
        print("M=%d N=%d K=%d stats=%d tune=%s rep %d: bad %d%s" % (M, N, K, want_stats, tune, rep, nb, msg), flush=True)
    ops.set_x3_tuning(0)

for tune in [(2,), (1,), (0,)]:
    for ws in (0, 1):
        run(9216, 512, 1024, ws, tune)
run(9216, 512, 512, 1, (2,))
run(4608, 512, 1024, 1, (2,))
