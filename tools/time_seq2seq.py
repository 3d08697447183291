"""Wall time of the seq2seq variant (configs A / D of BASELINE.json) on the GPU: one wireframe,
label_seq_length 259, default-xavier weights (never emits EOS early with seed 3: all 258 steps)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faceformer_amd.config import load_cfg  # noqa: E402
from faceformer_amd.models import SurfaceFormer  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402

for cfgfile, n in (("seq2seq.yml", 64), ("seq2seq+coedge.yml", 216)):
    cfg = load_cfg(os.path.join(ROOT, "configs", cfgfile))
    L, T = cfg.model.num_lines, cfg.model.label_seq_length
    model = SurfaceFormer(**cfg.model)
    model.load_state_dict(make_state_dict(state_dict_spec("seq2seq", L, T), "gain4", 0))
    model = model.eval().cuda()
    b = make_wireframes(n, L, T, "seq2seq", seeds=[3])
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = model(dict(b))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    steps = out["pointer"].shape[1]
    print("%s: n=%d steps=%d  %.1f ms  -> %.0f selections/s" % (cfgfile, n, steps, dt * 1e3, steps / dt))
