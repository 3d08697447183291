"""ORACLE TOOLING (build container only): run the reference dataset classes (np.int shim) on
synthetic wireframe JSONs and store raw inputs + produced items in tests/golden/data_cases.npz/.json."""
import json
import os
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def main():
    np.int = int
    np.bool = bool
    pkg = types.ModuleType("faceformer")
    pkg.__path__ = [os.path.join(REFERENCE, "faceformer")]
    sys.modules["faceformer"] = pkg
    from faceformer.datasets.data import ABCDataset
    from faceformer.datasets.data_para import ABCDataset_Parallel
    rng = np.random.default_rng(7)
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    cfg = types.SimpleNamespace(num_points_per_line=50, num_lines=24, point_dim=2, label_seq_length=80,
                                max_num_faces=42, max_face_length=14, token=tok)
    raws = []
    for _ in range(4):
        n = int(rng.integers(6, 20))
        edges = []
        for _e in range(n):
            k = 2 if rng.random() < 0.6 else int(rng.integers(3, 70))
            edges.append(rng.uniform(-1, 1, size=(k, 2)).round(6).tolist())
        faces_par, faces_seq = [], []
        for _f in range(int(rng.integers(1, 4))):
            loops = [rng.choice(n, size=int(rng.integers(2, 5)), replace=False).tolist()
                     for _l in range(int(rng.integers(1, 3)))]
            faces_par.append([int(rng.integers(0, 4)), loops])
            faces_seq.append(loops if rng.random() < 0.5 else loops[0])
        raws.append({"edges": edges, "par": faces_par, "seq": faces_seq})
    out = {"cfg": {"num_points_per_line": 50, "num_lines": 24, "point_dim": 2, "label_seq_length": 80,
                   "max_face_length": 14}, "cases": []}
    arrays = {}
    with tempfile.TemporaryDirectory() as d:
        for i, r in enumerate(raws):
            for kind, cls, key in (("par", ABCDataset_Parallel, "par"), ("seq", ABCDataset, "seq")):
                raw = {"edges": r["edges"], "faces_indices": r[key]}
                fn = "%s_%d.json" % (kind, i)
                with open(os.path.join(d, fn), "w") as f:
                    json.dump(raw, f)
                item = cls(d, fn, cfg)[0]
                tag = "%s_%d" % (kind, i)
                out["cases"].append({"tag": tag, "kind": kind, "raw": raw,
                                     "num_input": int(item["num_input"])})
                for k in ("input", "label", "input_mask", "label_mask"):
                    arrays[tag + "/" + k] = np.asarray(item[k])
    with open(os.path.join(ROOT, "tests", "golden", "data_cases.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "data_cases.npz"), **arrays)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
