"""Host-side time of one `model(batch)` call of config B before the first engine kernel is enqueued and after the last one:
wall-clock stamps at the entry of forward, at the ff_encode / ff_decode FFI calls and at the return (GPU idle in between:
every pass ends with ff_decode's own stream synchronisation).    python tools/host_preamble_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faceformer_amd.config import load_cfg  # noqa: E402
from faceformer_amd.hip import engine as eng_mod  # noqa: E402
from faceformer_amd.models import SurfaceFormer_Parallel  # noqa: E402
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec  # noqa: E402

cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"), ["model.num_lines", "256"])
T = cfg.model.max_face_length
model = SurfaceFormer_Parallel(**cfg.model)
model.load_state_dict(make_state_dict(state_dict_spec("parallel", 256, T), "default", 0))
model = model.eval().cuda()
model.x3_min_rows = 0
batch = make_wireframes([256], 256, T, "parallel", seeds=[0])
batch = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}
stamps = {}
lib = model.engine()._lib
orig_enc, orig_dec = eng_mod.PathEngine.encode, eng_mod.PathEngine.decode


def enc(self, *a, **k):
    stamps["encode_in"] = time.perf_counter()
    r = orig_enc(self, *a, **k)
    stamps["encode_out"] = time.perf_counter()
    return r


def dec(self, *a, **k):
    stamps["decode_in"] = time.perf_counter()
    r = orig_dec(self, *a, **k)
    stamps["decode_out"] = time.perf_counter()
    return r


eng_mod.PathEngine.encode, eng_mod.PathEngine.decode = enc, dec
for _ in range(3):
    with torch.no_grad():
        model(dict(batch))
torch.cuda.synchronize()
acc = {}
N = 20
for _ in range(N):
    t0 = time.perf_counter()
    with torch.no_grad():
        out = model(dict(batch))
    t1 = time.perf_counter()
    for k, v in (("forward entry -> PathEngine.encode", stamps["encode_in"] - t0), ("inside encode (host: kv_len, alloc, FFI enqueue)", stamps["encode_out"] - stamps["encode_in"]),
                 ("encode -> decode", stamps["decode_in"] - stamps["encode_out"]), ("inside decode (incl. the GPU run)", stamps["decode_out"] - stamps["decode_in"]),
                 ("decode return -> forward return", t1 - stamps["decode_out"]), ("whole call", t1 - t0)):
        acc[k] = acc.get(k, 0.0) + v
for k, v in acc.items():
    print("%-52s %9.1f us" % (k, v / N * 1e6))
