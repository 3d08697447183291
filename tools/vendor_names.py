"""Which vendor kernels does torch.mm pick for the f32 projection shapes of the path (run under rocprofv3 --kernel-trace)."""
import torch
for M in (4096, 9216, 16384, 32768):
    for (N, K) in ((1536, 512), (512, 512), (1024, 512), (512, 1024)):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda")
        for _ in range(3):
            torch.mm(a, w)
        torch.cuda.synchronize()
