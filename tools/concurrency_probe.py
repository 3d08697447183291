"""Does decoding independent sequence groups on concurrent HIP streams hide the per-kernel latency
floor?  Decode W wireframes (a) in one call, (b) one call per wireframe sequentially, (c) one host
thread + stream per wireframe."""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faceformer_amd.config import load_cfg  # noqa: E402
from faceformer_amd.models import SurfaceFormer_Parallel  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = 256
cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"), ["model.num_lines", str(n)])
T = cfg.model.max_face_length
spec = state_dict_spec("parallel", n, T)
sd = make_state_dict(spec, "default", 0)
dev = torch.device("cuda")


def mk():
    m = SurfaceFormer_Parallel(**cfg.model)
    m.load_state_dict(sd)
    return m.eval().to(dev)


models = [mk() for _ in range(W)]
batches = []
for i in range(W):
    b = make_wireframes(n, n, T, "parallel", seeds=[i])
    batches.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})
ball = make_wireframes(n, n, T, "parallel", seeds=list(range(W)))
ball = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in ball.items()}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def one_call():
    with torch.no_grad():
        models[0](dict(ball))


def sequential():
    with torch.no_grad():
        for i in range(W):
            models[i](dict(batches[i]))


streams = [torch.cuda.Stream() for _ in range(W)]


def threaded():
    def work(i):
        with torch.cuda.stream(streams[i]), torch.no_grad():
            models[i](dict(batches[i]))
            streams[i].synchronize()
    ts = [threading.Thread(target=work, args=(i,)) for i in range(W)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


print("W=%d  one batched call: %.1f ms | sequential calls: %.1f ms | %d threads+streams: %.1f ms"
      % (W, timed(one_call), timed(sequential), W, timed(threaded)))
