"""ORACLE TOOLING (build container only): end-to-end golden for the decode CLI with the co-edge
post-processing branch ON (the reference's default, faceformer/config.py:52).

For two synthetic wireframe JSONs (closed curves through a common point, a few open segments, co-edge
`pairings`) it runs the IMPORTED reference end to end on CPU --

    ABCDataset_Parallel.__getitem__        (faceformer/datasets/data_para.py:56-110, np.int shim)
    SurfaceFormer_Parallel.forward_eval    (faceformer/models/model_para.py:181-241)
    Trainer.face_accuracy                  (faceformer/trainer.py:210-300: parse_parallel_faces,
                                            filter_faces_by_encloseness, map_coedge_into_edges, majority vote)

-- and stores the raw wireframes, the weight recipe and the records the reference's test_step would dump
(`pred_faces`, `label_faces`; trainer.py:118-136) in tests/golden/cli_coedge_case.json.  Only data is
stored; nothing of the reference is copied.  The GPU test replays the same files through main.py.
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"
sys.path.insert(0, ROOT)

MODEL = dict(num_model=128, num_head=2, num_feedforward=256, num_encoder_layers=2, num_decoder_layers=2,
             num_lines=12, max_face_length=8)
RECIPE, WSEED = "gain4", 21


def closed_curve(rng, p, k):
    """k-point polyline that starts and ends EXACTLY at p (a loop on its own for is_face_enclosed)."""
    c = p + rng.uniform(-0.6, 0.6, size=2)
    r = np.linalg.norm(c - p)
    a0 = np.arctan2(p[1] - c[1], p[0] - c[0])
    ang = a0 + np.linspace(0, 2 * np.pi, k)
    pts = c + r * np.stack([np.cos(ang), np.sin(ang)], 1)
    pts[0] = p
    pts[-1] = p
    return pts.round(6).tolist()


def make_raw(seed):
    rng = np.random.default_rng([0xC11, seed])
    p = np.array([0.125, -0.25])
    edges = []
    for i in range(12):
        if i in (4, 9, 11):   # open segments: faces that use them do not close
            edges.append([p.tolist(), rng.uniform(-1, 1, size=2).round(6).tolist()])
        else:
            edges.append(closed_curve(rng, p, int(rng.integers(5, 40))))
    faces = [[0, [[0, 1]]], [1, [[2, 3, 5]]], [3, [[6], [8, 7]]], [0, [[4, 0]]]]
    return {"edges": edges, "faces_indices": faces, "pairings": {"3": 1, "7": 2, "10": 0},
            "dominant_directions": [[1, 0, 0], [0, 1, 0], [0, 0, 1]]}


def import_reference():
    np.int = int
    np.bool = bool
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = type("LightningModule", (), {})
    pl.Callback = type("Callback", (), {})
    sys.modules["pytorch_lightning"] = pl
    npe = types.ModuleType("numpyencoder")
    npe.NumpyEncoder = json.JSONEncoder
    sys.modules["numpyencoder"] = npe
    pkg = types.ModuleType("faceformer")
    pkg.__path__ = [os.path.join(REFERENCE, "faceformer")]
    sys.modules["faceformer"] = pkg
    sys.path.insert(0, REFERENCE)
    import faceformer.models as ref_models
    import faceformer.trainer as tr
    from faceformer.datasets.data_para import ABCDataset_Parallel
    return ref_models, tr, ABCDataset_Parallel


def jsonable(x):
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    return x


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("needs /root/reference (build container only)")
    ref_models, tr, DS = import_reference()
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    m = MODEL
    model = ref_models.SurfaceFormer_Parallel(
        num_model=m["num_model"], num_head=m["num_head"], num_feedforward=m["num_feedforward"],
        num_encoder_layers=m["num_encoder_layers"], num_decoder_layers=m["num_decoder_layers"], dropout=0.2,
        num_lines=m["num_lines"], max_face_length=m["max_face_length"], token=tok).eval()
    spec = state_dict_spec("parallel", m["num_lines"], m["max_face_length"], m["num_model"], m["num_feedforward"],
                           m["num_encoder_layers"], m["num_decoder_layers"])
    model.load_state_dict(make_state_dict(spec, RECIPE, WSEED))
    dcfg = types.SimpleNamespace(num_points_per_line=50, num_lines=m["num_lines"], point_dim=2, max_num_faces=42,
                                 max_face_length=m["max_face_length"], token=tok)
    raws = [make_raw(s) for s in (1, 2)]
    out = {"model": m, "recipe": RECIPE, "wseed": WSEED, "tol": 2e-4, "samples": []}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "json"))
        names = []
        for i, raw in enumerate(raws):
            names.append("json/%08d.json" % i)
            with open(os.path.join(d, names[-1]), "w") as f:
                json.dump(raw, f)
        with open(os.path.join(d, "test.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        ds = DS(d, ["test.txt"], dcfg)
        hp = types.SimpleNamespace(model=types.SimpleNamespace(token=tok),
                                   post_process=types.SimpleNamespace(is_coedge=True, enclosedness_tol=2e-4))
        me = types.SimpleNamespace(hparams=hp, dataset=ds)
        me.parse_parallel_faces = lambda *a: tr.Trainer.parse_parallel_faces(me, *a)
        me.parse_faces = lambda *a: tr.Trainer.parse_faces(me, *a)
        for i in range(len(ds)):
            item = ds[i]
            batch = {"input": torch.from_numpy(item["input"])[None], "input_mask": torch.from_numpy(item["input_mask"])[None],
                     "label": torch.from_numpy(item["label"])[None], "num_input": [item["num_input"]], "id": [i]}
            margins = []
            orig = torch.argmax

            def spy(x, *a, **k):
                v = torch.sort(x.squeeze(-1), dim=1, descending=True).values
                margins.append(float((v[:, 0] - v[:, 1]).min()))
                return orig(x, *a, **k)
            torch.argmax = spy
            try:
                with torch.no_grad():
                    outputs = model(batch)
            finally:
                torch.argmax = orig
            predict = outputs["predict"][0].numpy().copy()   # face_accuracy edits `predict` in place (trainer.py:181-208)
            _, outputs = tr.Trainer.face_accuracy(me, outputs)
            out["samples"].append({
                "raw": raws[i], "predict": predict,
                "pred_faces": outputs["predictions"][0], "label_faces": outputs["labels"][0],
                "precision": outputs["precisions"][0], "recall": outputs["recalls"][0],
                "min_margin": min(margins)})
            print("sample", i, "pred_faces", len(outputs["predictions"][0]), "label_faces", len(outputs["labels"][0]),
                  "precision", outputs["precisions"][0], "min margin %.4g" % min(margins))
    path = os.path.join(ROOT, "tests", "golden", "cli_coedge_case.json")
    with open(path, "w") as f:
        json.dump(jsonable(out), f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
