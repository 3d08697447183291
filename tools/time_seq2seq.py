"""Wall time of the seq2seq variant (configs A / D of BASELINE.json) on the GPU: one wireframe per call (the
configurations as BASELINE states them) and batches of 8 / 64 wireframes per call (the reference's forward_eval
takes a batch; its stop rule waits for every wireframe's EOS), label_seq_length 259."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faceformer_amd.config import load_cfg  # noqa: E402
from faceformer_amd.models import SurfaceFormer  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402

CASES = (("seq2seq.yml", 64), ("seq2seq+coedge.yml", 216))
BATCHES = tuple(int(v) for v in os.environ.get("FF_SEQ_BATCHES", "1,8,64").split(","))
if os.environ.get("FF_SEQ_ONLY_A"):   # kernel-trace runs: config A, one wireframe per call
    CASES = CASES[:1]
    if "FF_SEQ_BATCHES" not in os.environ:
        BATCHES = (1,)
for cfgfile, n in CASES:
    cfg = load_cfg(os.path.join(ROOT, "configs", cfgfile))
    L, T = cfg.model.num_lines, cfg.model.label_seq_length
    model = SurfaceFormer(**cfg.model)
    model.load_state_dict(make_state_dict(state_dict_spec("seq2seq", L, T), "gain4", 0))
    model = model.eval().cuda()
    if os.environ.get("FF_SEQ_X3_MIN_ROWS"):   # 0 = every product on the f32 matrix cores (default: the package default)
        model.x3_min_rows = int(os.environ["FF_SEQ_X3_MIN_ROWS"])
    if os.environ.get("FF_SEQ_CHUNK"):   # sequences per micro-batch (default: the model's 256)
        model.chunk_max_seqs = int(os.environ["FF_SEQ_CHUNK"])
    for nb in BATCHES:
        b = make_wireframes(n, L, T, "seq2seq", seeds=list(range(3, 3 + nb)))
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        best = 1e9
        for rep in range(int(os.environ.get("FF_SEQ_REPS", "5"))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                out = model(dict(b))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        steps = out["pointer"].shape[1]
        print("%s: n=%d wireframes/call=%d steps=%d  %.1f ms  -> %.0f selections/s (%.2f ms per wireframe)"
              % (cfgfile, n, nb, steps, best * 1e3, nb * steps / best, best * 1e3 / nb))
