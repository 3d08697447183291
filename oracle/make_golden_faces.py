"""ORACLE TOOLING (build container only): capture input/output pairs of the reference's face parsing
and post-processing functions into tests/golden/faces_cases.json.

The reference harness (faceformer/trainer.py) needs pytorch_lightning / numpyencoder, which are not
installed: both are replaced by empty stub modules for the import, and the methods are called unbound
on a SimpleNamespace carrying `hparams` (exactly as SURVEY.md 8c describes).  Nothing of the
reference is copied; only data (inputs and the outputs it produced) is stored.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = "/root/reference"


def import_reference():
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
    sys.path.insert(0, REFERENCE)  # for `dataset.tests.check_faces_enclosed`
    import faceformer.trainer as tr
    import faceformer.post_processing as pp
    return tr, pp


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
    if isinstance(x, bool):
        return bool(x)
    return x


def polygon_wireframe(rng, n_loops, tol_scale=1.0):
    """Closed polygons (each edge a 2-point segment, head-to-tail) + a few stray edges."""
    edges, loops = [], []
    for _ in range(n_loops):
        k = int(rng.integers(3, 7))
        c = rng.uniform(-1, 1, size=2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, size=k))
        pts = c + 0.3 * np.stack([np.cos(ang), np.sin(ang)], 1)
        loop = []
        for i in range(k):
            loop.append(len(edges))
            edges.append([pts[i].tolist(), pts[(i + 1) % k].tolist()])
        loops.append(loop)
    for _ in range(3):
        edges.append(rng.uniform(-1, 1, size=(2, 2)).tolist())
    return edges, loops


def main():
    tr, pp = import_reference()
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    hp = types.SimpleNamespace(model=types.SimpleNamespace(token=tok))
    me = types.SimpleNamespace(hparams=hp)
    rng = np.random.default_rng(123)
    cases = {"parallel": [], "seq": [], "enclosed": [], "filter": [], "coedge": []}

    # --- parse_parallel_faces: real decoder outputs + random rows with terminators ---
    for name in ("par_small_gain4", "par_small_ragged", "par_small_earlybreak", "par_full_n40_gain4"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        case = json.loads(bytes(z["case"]).decode())
        for w, n in enumerate(case["n_edges"]):
            pred = z["predict"][w].astype(np.int64)
            lab = rng.integers(0, n + 4, size=(max(1, n // 2), pred.shape[1])).astype(np.int64)
            for r in lab:
                r[int(rng.integers(1, len(r))):] = 0
                r[int(rng.integers(1, len(r)))] = int(rng.integers(1, 4))
            pf, lf = tr.Trainer.parse_parallel_faces(me, pred.copy(), lab.copy(), n)
            cases["parallel"].append({"predicts": pred, "labels": lab, "num_edges": n, "pred_faces": pf, "label_faces": lf})
    for _ in range(8):
        n, T, F = int(rng.integers(5, 40)), int(rng.integers(4, 20)), int(rng.integers(1, 30))
        pred = rng.integers(0, n + 10, size=(F, T)).astype(np.int64)
        pred[rng.random((F, T)) < 0.15] = rng.integers(0, 4)
        lab = rng.integers(0, n + 4, size=(F, T)).astype(np.int64)
        lab[rng.random((F, T)) < 0.2] = rng.integers(1, 4)
        pf, lf = tr.Trainer.parse_parallel_faces(me, pred.copy(), lab.copy(), n)
        cases["parallel"].append({"predicts": pred, "labels": lab, "num_edges": n, "pred_faces": pf, "label_faces": lf})

    # --- parse_faces (single sequence) ---
    for name in ("seq_small_default", "seq_small_gain4", "seq_small_eos", "seq_full_A64_gain4"):
        z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        case = json.loads(bytes(z["case"]).decode())
        for w, n in enumerate(case["n_edges"]):
            pred = z["predict"][w].astype(np.int64)
            lab = rng.integers(2, n + 4, size=pred.shape[0]).astype(np.int64)
            lab[int(rng.integers(3, len(lab)))] = 3
            pf, lf = tr.Trainer.parse_faces(me, pred.copy(), lab.copy(), n)
            cases["seq"].append({"predicts": pred, "labels": lab, "num_edges": n, "pred_faces": pf, "label_faces": lf})
    for _ in range(8):
        n, T = int(rng.integers(5, 40)), int(rng.integers(6, 60))
        pred = rng.integers(0, n + 10, size=T).astype(np.int64)
        pred[rng.random(T) < 0.2] = 2
        if rng.random() < 0.7:
            pred[int(rng.integers(1, T))] = 3
        lab = rng.integers(2, n + 6, size=T).astype(np.int64)
        lab[rng.random(T) < 0.2] = 2
        lab[int(rng.integers(1, T))] = 3
        pf, lf = tr.Trainer.parse_faces(me, pred.copy(), lab.copy(), n)
        cases["seq"].append({"predicts": pred, "labels": lab, "num_edges": n, "pred_faces": pf, "label_faces": lf})

    # --- is_face_enclosed / filter_faces_by_encloseness / map_coedge_into_edges ---
    from dataset.tests.check_faces_enclosed import is_face_enclosed
    tol = 2e-4
    for _ in range(6):
        edges, loops = polygon_wireframe(rng, int(rng.integers(1, 4)))
        faces = []
        for lp in loops:
            k = int(rng.integers(0, len(lp)))
            faces.append((int(rng.integers(0, 3)), tuple(lp[k:] + lp[:k])))          # rotated closed loop
        if len(loops) >= 2:
            faces.append((1, tuple(loops[1] + loops[0])))                               # two loops in one face
        faces.append((0, tuple(loops[0][:-1])))                                         # open chain
        faces.append((2, tuple(loops[0][::-1])))                                        # wrong direction
        faces.append((0, tuple(list(loops[0]) + [len(edges) + 5])))                     # out-of-range index skipped
        for _, f in faces:
            cases["enclosed"].append({"edges": edges, "face": list(f), "tol": tol,
                                      "result": is_face_enclosed(edges, f, tol)})
        cases["filter"].append({"edges": edges, "faces": faces, "tol": tol,
                                "result": pp.filter_faces_by_encloseness(edges, faces, tol)})
    pairings = {"3": 1, "7": 2, "10": 0}
    for _ in range(4):
        idx = rng.integers(0, 12, size=int(rng.integers(1, 9))).tolist()
        cases["coedge"].append({"pairings": pairings, "indices": idx,
                                "result": pp.map_coedge_into_edges(pairings, idx)})

    out = os.path.join(ROOT, "tests", "golden", "faces_cases.json")
    with open(out, "w") as f:
        json.dump(jsonable(cases), f)
    print("wrote", out, {k: len(v) for k, v in cases.items()}, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
