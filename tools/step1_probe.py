"""Un-traced wall time of the FIRST decode step of config B (encoder + per-batch invariants + step 1 + finalize): the same model
decoded with max_face_length T = 2 ... 9 (1 ... 8 steps), device-synchronised, median of 30 calls each.  The intercept of the
line through the points is what a call costs before its second step; profiles/r05/steps.txt shows 1.8-2.5 ms for it UNDER the
tracer (gaps of 40-210 us in front of 5-10 us kernels, i.e. the tracer's per-launch host cost), this measures it without.
    python tools/step1_probe.py            -> profiles/r06/step1_untraced.txt
"""
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.models import SurfaceFormer_Parallel  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402


def main():
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    L = 256
    rows = []
    for T in (2, 3, 4, 5, 7, 9, 37):
        model = SurfaceFormer_Parallel(num_model=512, num_head=8, num_feedforward=1024, num_encoder_layers=6, num_decoder_layers=6,
                                       dropout=0.2, num_lines=L, max_face_length=T, token=tok)
        sd = make_state_dict(state_dict_spec("parallel", L, T, 512, 1024, 6, 6), "default", 0)
        model.load_state_dict(sd)
        model = model.eval().cuda()
        model.x3_min_rows = 0          # the f32 headline form
        b = make_wireframes([L], L, T, "parallel", seeds=[0])
        b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
        ts = []
        with torch.no_grad():
            for i in range(36):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                model(dict(b))
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        ts = np.sort(np.array(ts[6:]))
        rows.append((T - 1, float(np.median(ts)), float(ts[0])))
        print("steps %2d: median %.3f ms  min %.3f ms" % rows[-1])
    s1 = rows[0][1]
    print("one-step call (encoder + invariants + step 1 + finalize + host): %.3f ms; each further small step adds %.3f ms "
          "(steps 2..4 mean); a 36-step call %.2f ms" % (s1, (rows[3][1] - rows[0][1]) / 3.0, rows[-1][1]))


if __name__ == "__main__":
    main()
