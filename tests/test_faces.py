"""CPU: face parsing / post-processing / JSON record against input-output pairs
captured from the imported reference (oracle/make_golden_faces.py)."""
import json
import os
import types

import numpy as np
import pytest

from conftest import GOLDEN, token_ns
from faceformer_amd import faces as F


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLDEN, "faces_cases.json")) as f:
        return json.load(f)


def as_faces(x):
    return [(int(t), tuple(int(i) for i in idx)) for t, idx in x]


def test_parse_parallel_faces_matches_reference(cases):
    tok = token_ns()
    assert len(cases["parallel"]) >= 10
    for c in cases["parallel"]:
        pred, lab = np.asarray(c["predicts"]), np.asarray(c["labels"])
        p0, l0 = pred.copy(), lab.copy()
        pf, lf = F.parse_parallel_faces(pred, lab, c["num_edges"], tok)
        assert pf == as_faces(c["pred_faces"])
        assert lf == as_faces(c["label_faces"])
        assert np.array_equal(pred, p0) and np.array_equal(lab, l0)   # inputs are not mutated


def test_parse_faces_matches_reference(cases):
    tok = token_ns()
    for c in cases["seq"]:
        pf, lf = F.parse_faces(np.asarray(c["predicts"]), np.asarray(c["labels"]), c["num_edges"], tok)
        assert pf == as_faces(c["pred_faces"])
        assert lf == as_faces(c["label_faces"])


def test_token_namespace_variants():
    """token may be a namespace, a CfgNode or a dict."""
    from faceformer_amd.config import load_cfg
    cfg = load_cfg("")
    rows = np.array([[5, 6, 2, 9], [4, 4, 4, 4], [3, 0, 0, 0]])
    a = F.parse_parallel_faces(rows, rows, 10, token_ns())
    b = F.parse_parallel_faces(rows, rows, 10, cfg.model.token)
    c = F.parse_parallel_faces(rows, rows, 10, dict(cfg.model.token))
    assert a == b == c
    assert a[0] == [(1, (1, 2)), (3, (0, 0, 0, 0))]   # unterminated row: type = last token - offset


def test_is_face_enclosed_and_filter_match_reference(cases):
    for c in cases["enclosed"]:
        got = F.is_face_enclosed(c["edges"], c["face"], c["tol"])
        want = c["result"]
        assert (got is False and want is False) or [list(lp) for lp in got] == want
    for c in cases["filter"]:
        faces = [(t, tuple(f)) for t, f in c["faces"]]
        got = F.filter_faces_by_encloseness(c["edges"], faces, c["tol"])
        want = [(t, tuple(tuple(lp) for lp in loops)) for t, loops in c["result"]]
        assert got == want
        assert len(got) >= 1
    for c in cases["coedge"]:
        assert F.map_coedge_into_edges(c["pairings"], c["indices"]) == c["result"]


def test_oriented_edges_and_coedge_filter():
    edges = [[[0, 0], [1, 0]], [[1, 1], [1, 0]], [[1, 1], [0, 0]]]
    assert F.is_face_enclosed(edges, [(0, 0), (1, 1), (2, 0)], 1e-6) == [[(0, 0), (1, 1), (2, 0)]]
    assert F.is_face_enclosed(edges, [0, 1, 2], 1e-6) is False
    faces = [(0, ((0, 1, 2),)), (1, ((3, 4),)), (0, ((5,),))]
    kept = F.filter_faces_by_coedge({3: 0, 5: 9}, faces)
    assert kept == [faces[0], faces[2]]       # face 1 reuses edge 0 through its co-edge 3


def test_metrics_majority_vote_and_json_roundtrip():
    pred = [(0, (3, 1, 2)), (1, (1, 2, 3)), (1, (2, 3, 1)), (2, (7, 8))]
    lab = [(1, (1, 2, 3)), (0, (4, 5, 6))]
    m = F.face_metrics(pred, lab)
    assert m["predictions"] == [(1, (1, 2, 3)), (2, (7, 8))]
    assert m["precision"] == 0.5 and m["recall"] == 0.5 and m["type_acc"] == 1.0
    assert F.face_metrics([], lab)["precision"] == 0
    rec = F.faces_record(np.zeros((2, 2, 2)), [[1.0, 0.0]], m["predictions"], m["labels"])
    back = json.loads(F.dumps_record(rec))
    assert set(back) == {"edges", "dominant_directions", "pred_faces", "label_faces"}
    assert back["pred_faces"] == [[1, [1, 2, 3]], [2, [7, 8]]]


def test_postprocess_pipeline():
    sq = [[[0, 0], [1, 0]], [[1, 0], [1, 1]], [[1, 1], [0, 1]], [[0, 1], [0, 0]]]
    out = F.postprocess_faces([(0, (2, 3, 0, 1)), (1, (0, 1))], sq, {"3": 7}, 2e-4)
    assert out == [(0, [0, 1, 2, 7])]


def test_coedge_branch_end_to_end_on_reference_tokens():
    """Host side of the co-edge CLI golden (oracle/make_golden_cli.py): from the reference's `predict` tokens of
    two wireframes with closed loops and pairings, parse -> enclosure filter -> co-edge mapping -> majority
    vote must give the reference's dumped `pred_faces` / `label_faces` and precision / recall."""
    import json
    import os
    import types
    from conftest import GOLDEN
    from faceformer_amd import datasets as D
    from faceformer_amd import faces as FZ
    gold = json.load(open(os.path.join(GOLDEN, "cli_coedge_case.json")))
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    m = gold["model"]
    for smp in gold["samples"]:
        raw = smp["raw"]
        cfgm = types.SimpleNamespace(num_points_per_line=50, num_lines=m["num_lines"], point_dim=2, max_num_faces=42,
                                     max_face_length=m["max_face_length"], label_seq_length=0, token=tok)
        item = D.pack_parallel(raw, cfgm) if hasattr(D, "pack_parallel") else None
        import numpy as np
        pred = np.asarray(smp["predict"], dtype=np.int64)
        if item is not None:
            lab = item["label"]
        else:
            import tempfile
            with tempfile.TemporaryDirectory() as d:
                json.dump(raw, open(os.path.join(d, "a.json"), "w"))
                lab = D.ABCDataset_Parallel(d, "a.json", cfgm)[0]["label"]
        pf, lf = FZ.parse_parallel_faces(pred, lab, len(raw["edges"]), tok)
        pf = FZ.postprocess_faces(pf, raw["edges"], raw["pairings"], gold["tol"])
        lf = FZ.postprocess_faces(lf, raw["edges"], raw["pairings"], gold["tol"])
        met = FZ.face_metrics(pf, lf)
        assert [[t, list(f)] for t, f in met["predictions"]] == smp["pred_faces"]
        assert sorted([t, list(f)] for t, f in met["labels"]) == sorted(smp["label_faces"])
        assert met["precision"] == smp["precision"] and met["recall"] == smp["recall"]


def test_own_stop_rule_restores_the_one_sample_tokens():
    """In a batch the reference's loop runs until every wireframe is done; a one-sample decode stops at the wireframe's own
    rule.  apply_own_stop_rule cuts a wireframe's rows back to what the one-sample decode leaves (main.py --batch-size N,
    dist.decode_to_face_json): parallel = the first step at which none of its sequences selects an edge, seq2seq = its EOS."""
    import types
    tok = types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    rows = np.array([[5, 6, 2, 9, 1, 0], [6, 7, 0, 8, 2, 0]])          # step 2: both below 4 -> the own loop stops there
    FZ = F
    cut = FZ.apply_own_stop_rule(rows, tok, True)
    assert cut.tolist() == [[5, 6, 2, 0, 0, 0], [6, 7, 0, 0, 0, 0]] and rows[0, 3] == 9     # a copy
    keep = np.array([[5, 6, 7, 2, 9, 9], [6, 7, 0, 8, 1, 0]])          # never all below 4 in one step: untouched
    assert FZ.apply_own_stop_rule(keep, tok, True).tolist() == keep.tolist()
    assert FZ.apply_own_stop_rule(np.array([1, 7, 8, 3, 9, 3, 0]), tok, False).tolist() == [1, 7, 8, 3, 0, 0, 0]
    # the sequence that had not terminated at the own stop step parses to the same (typeless) face either way
    a, _ = FZ.parse_parallel_faces(cut, cut, 8, tok)
    b, _ = FZ.parse_parallel_faces(FZ.apply_own_stop_rule(cut, tok, True), cut, 8, tok)
    assert a == b
