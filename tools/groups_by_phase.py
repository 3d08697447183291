"""Do two sequence groups on two streams pay at LARGE t?  (run on the GPU box)
    python tools/groups_by_phase.py > profiles/r05/groups_by_phase.txt

Round 3 measured ONE config-B wireframe as 2 x 128 sequences on two HIP streams over the whole decode: 65.7 against 61.8 ms
(`sequence_groups_on_streams.txt`) -- a latency-bound small-t step costs the same for 128 sequences as for 256, so the groups pay
it twice.  At large t the launches are 30-120 us and ~5 us of each is launch boundary, which a second stream's blocks could fill.
This probe separates the two phases: the same model with max_face_length 8 / 16 / 24 / 36 (the decode's first 8 / 16 / 24 / 36 steps),
one group against two groups on two streams; the large-t phase of a form = its time at 36 minus its time at the shorter length."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faceformer_amd.config import load_cfg  # noqa: E402
from faceformer_amd.models import SurfaceFormer_Parallel  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402

REPS = int(os.environ.get("FF_REPS", "12"))
rows = {}
for T in (8, 16, 24, 36):
    cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"), ["model.num_lines", "256", "model.max_face_length", str(T)])
    model = SurfaceFormer_Parallel(**cfg.model)
    spec = state_dict_spec("parallel", 256, T, cfg.model.num_model, cfg.model.num_feedforward,
                           cfg.model.num_encoder_layers, cfg.model.num_decoder_layers)
    model.load_state_dict(make_state_dict(spec, "default", 0))
    model = model.eval().cuda()
    model.x3_min_rows = 0
    b = make_wireframes([256], 256, T, "parallel", seeds=[0])
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    for name, cseq, nstr in (("1 group", 0, 1), ("2 x 128 on 2 streams", 128, 2), ("4 x 64 on 4 streams", 64, 4)):
        model.chunk_seqs, model.num_streams = cseq, nstr
        ts = []
        for rep in range(REPS + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.no_grad():
                out = model(dict(b))
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        rows[(T, name)] = (1e3 * ts[0], 1e3 * ts[len(ts) // 2])
        print("max_face_length %2d  %-22s best %7.3f ms  median %7.3f ms" % (T, name, rows[(T, name)][0], rows[(T, name)][1]))
        sys.stdout.flush()
print()
for name in ("1 group", "2 x 128 on 2 streams", "4 x 64 on 4 streams"):
    for T0 in (8, 16, 24):
        print("%-22s steps %2d..36: %7.3f ms (medians)" % (name, T0 + 1, rows[(36, name)][1] - rows[(T0, name)][1]))
