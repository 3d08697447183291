"""One ff_gemm_x3 / ff_gemm_f32 launch shape for timing and rocprofv3 --pmc passes.
    python tools/x3_probe.py M N K [iters] [f32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import ops  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
mode = sys.argv[5] if len(sys.argv) > 5 else "x3"   # x3 | x3p (pre-split activations) | f32
f32 = mode == "f32"
a = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * 0.05
out = torch.empty(M, N, device="cuda")
planes = ops.split_weight(w)
ap = ops.split_weight(a)
if f32:
    fn = lambda: ops.linear(a, w, None, out=out)
elif mode == "x3p":
    fn = lambda: ops.linear_x3(None, planes, None, out=out, x_planes=ap)
else:
    fn = lambda: ops.linear_x3(a, planes, None, out=out)
fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    fn()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print("M=%d N=%d K=%d %s: %.1f us, %.1f TF/s" % (M, N, K, mode, us, 2.0 * M * N * K / us / 1e6))
