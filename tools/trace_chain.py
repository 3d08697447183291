import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
from faceformer_amd.config import load_cfg
from faceformer_amd.models import SurfaceFormer
from faceformer_amd.hip import lib as L
from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec
cfg = load_cfg("configs/seq2seq.yml")
Ln, T = cfg.model.num_lines, cfg.model.label_seq_length
m = SurfaceFormer(**cfg.model)
m.load_state_dict(make_state_dict(state_dict_spec("seq2seq", Ln, T), "gain4", 0))
m = m.eval().cuda()
m.decode_flags |= L.FF_CHAIN
b = make_wireframes([64], Ln, T, "seq2seq", seeds=[3])
b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
with torch.no_grad():
    m(dict(b))
