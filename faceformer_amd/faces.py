"""From decoded token tensors to face loops and the per-wireframe JSON record.

Host-side integer / list logic that follows the decode path in the reference harness:

* `parse_parallel_faces` / `parse_faces`          reference faceformer/trainer.py:181-208 / 153-177
* `unique_faces_with_majority_type`, `face_metrics` reference trainer.py:257-293 (precision / recall /
  type accuracy of de-duplicated faces; majority vote over the types predicted for the same edge set)
* `is_face_enclosed`                               reference dataset/tests/check_faces_enclosed.py:11-46
* `filter_faces_by_encloseness`, `map_coedge_into_edges`, `filter_faces_by_coedge`
                                                   reference faceformer/post_processing.py:8-48
* `faces_record` / `dumps_record`                  the JSON written per sample, trainer.py:118-136
  (`edges`, `dominant_directions`, `pred_faces`, `label_faces`)

Unlike the reference, nothing here mutates its inputs (the reference subtracts the token offset in
place on views of `predict`).  Pinned by goldens captured from the imported reference
(`oracle/make_golden_faces.py` -> `tests/golden/faces_*.json`).
"""
import json
from collections import Counter

import numpy as np

__all__ = ["parse_parallel_faces", "parse_faces", "unique_faces_with_majority_type", "face_metrics",
           "is_face_enclosed", "filter_faces_by_encloseness", "map_coedge_into_edges",
           "filter_faces_by_coedge", "postprocess_faces", "faces_record", "dumps_record"]


def _tok(token, name, default):
    try:
        return int(getattr(token, name))
    except AttributeError:
        try:
            return int(token[name])
        except (KeyError, TypeError):
            return default


def _prefix_through_first(row, hit):
    """row[: first index where hit is True, inclusive]; the whole row if there is no hit."""
    idx = np.flatnonzero(hit)
    return row if idx.size == 0 else row[: idx[0] + 1]


def _edge_indices(tokens, offset, num_edges=None):
    v = np.asarray(tokens, dtype=np.int64) - offset
    v = v[v >= 0]
    if num_edges is not None:
        v = v[v < num_edges]
    return tuple(int(x) for x in v)


def _parallel_rows(rows, token, num_edges):
    off, ntok = _tok(token, "face_type_offset", 1), _tok(token, "len", 4)
    faces = []
    for row in np.asarray(rows):
        row = np.asarray(row, dtype=np.int64)
        seq = _prefix_through_first(row, (row >= off) & (row < ntok))
        if seq.size == 0:
            continue
        face_type = int(seq[-1]) - off
        idx = _edge_indices(seq, ntok, num_edges)
        if idx:
            faces.append((face_type, idx))
    return faces


def apply_own_stop_rule(predict, token, parallel, eos=None):
    """Tokens of ONE wireframe as a one-sample decode leaves them: zero behind the step at which the reference's loop would
    have stopped had the batch held only this wireframe.  In a batch the loop runs until EVERY wireframe is done (parallel
    model: the first step at which no sequence of the batch selects an edge, model_para.py:232; seq2seq: all EOS emitted,
    model.py:207-210), so a sequence that had not produced its terminator when its own wireframe's rule fired keeps decoding and
    may still produce one -- the one-sample decode (the reference's test loader, trainer.py:51) never sees those tokens.
    predict: [F, T] (parallel: the wireframe's own anchor rows) or [T] (seq2seq).  Returns a copy."""
    p = np.array(predict, dtype=np.int64, copy=True)
    ntok = _tok(token, "len", 4)
    if parallel:
        rows = p.reshape(-1, p.shape[-1])
        for j in range(1, rows.shape[1]):
            if (rows[:, j] < ntok).all():
                rows[:, j + 1:] = 0
                break
        return rows.reshape(p.shape)
    e = _tok(token, "EOS", 3) if eos is None else eos
    hit = np.nonzero(p[1:] == e)[0]
    if hit.size:
        p[hit[0] + 2:] = 0
    return p


def parse_parallel_faces(predicts, labels, num_edges, token):
    """One face per row: tokens up to and including the first face-type token (a special token in
    [face_type_offset, len)); the face type is that token minus the offset; edge indices are the
    tokens minus `len`, negatives dropped, and -- for predictions only -- indices >= num_edges dropped.
    Returns (predict_faces, label_faces) as lists of (type, (edge, ...))."""
    return _parallel_rows(predicts, token, num_edges), _parallel_rows(labels, token, None)


def _seq_faces(seq, token, num_edges, skip_single):
    eos, sep, ntok = _tok(token, "EOS", 3), _tok(token, "SEP", 2), _tok(token, "len", 4)
    seq = np.asarray(seq, dtype=np.int64)
    seq = _prefix_through_first(seq, seq == eos)
    cuts = np.flatnonzero(seq == sep) + 1
    faces = []
    for piece in np.split(seq, cuts):
        if skip_single and piece.size <= 1:
            continue
        idx = _edge_indices(piece[:-1], ntok, num_edges)   # the last token of a piece is its separator
        if idx:
            faces.append((0, idx))
    return faces


def parse_faces(predicts, labels, num_edges, token):
    """Single-sequence variant: cut after the first EOS, split after every SEP, drop the last token of
    every piece, subtract `len`, keep 0 <= index < num_edges.  Faces are typed 0."""
    return _seq_faces(predicts, token, num_edges, True), _seq_faces(labels, token, num_edges, False)


def unique_faces_with_majority_type(faces):
    """Group faces by their SET of edge indices; type = most common predicted type (first seen wins
    ties).  Returns [(type, sorted_unique_edges)] in first-seen order."""
    votes = {}
    for ftype, idx in faces:
        key = tuple(sorted(set(idx)))
        votes.setdefault(key, []).append(ftype)
    return [(Counter(types).most_common(1)[0][0], key) for key, types in votes.items()]


def face_metrics(predict_faces, label_faces):
    """precision, recall, type accuracy over de-duplicated faces (reference trainer.py:257-293).
    Also returns the two de-duplicated lists."""
    label_set = list(set((ftype, tuple(sorted(set(idx)))) for ftype, idx in label_faces))
    pred_set = unique_faces_with_majority_type(predict_faces)
    face_tp = type_tp = 0
    for ptype, pface in pred_set:
        for ltype, lface in label_set:
            if pface == lface:
                face_tp += 1
                type_tp += int(ptype == ltype)
                break
    if not pred_set or not label_set:
        prec = rec = tacc = 0
    else:
        prec, rec = face_tp / len(pred_set), face_tp / len(label_set)
        tacc = type_tp / face_tp if face_tp else 0
    return {"precision": prec, "recall": rec, "type_acc": tacc, "predictions": pred_set, "labels": label_set}


# ---- geometric post-processing (co-edge configs) ---------------------------------------------------
def _connects(e1, e2, tol):
    return abs(e1[-1][0] - e2[0][0]) < tol and abs(e1[-1][1] - e2[0][1]) < tol


def is_face_enclosed(edges, face_indices, tol):
    """Walk the face's edges in order; every edge must start where the previous one ended, and a loop
    closes when an edge ends at the loop's first start point.  Items may be `(index, reversed)` pairs.
    Returns the list of loops (lists of the original items) or False."""
    loops, current = [], []
    first = prev = None
    for item in face_indices:
        if isinstance(item, tuple):
            i, rev = item
            edge = edges[i][::-1] if rev else edges[i]
        elif item < len(edges):
            edge = edges[item]
        else:
            continue
        if first is None:
            first = edge
        elif not _connects(prev, edge, tol):
            return False
        prev = edge
        current.append(item)
        if _connects(edge, first, tol):
            loops.append(current)
            current, first = [], None
    return loops if first is None else False


def filter_faces_by_encloseness(edges, faces, tol):
    """Keep faces whose edges chain into closed loops; each loop is rotated so its smallest index comes
    first, loops are ordered by that index.  [(type, (edges...))] -> [(type, ((loop...), ...))]."""
    kept = []
    for ftype, face in faces:
        loops = is_face_enclosed(edges, face, tol)
        if not loops:
            continue
        rolled = []
        for loop in loops:
            arr = np.asarray(loop)
            rolled.append(tuple(np.roll(arr, -int(np.argmin(arr)), axis=0).astype(int).tolist()))
        kept.append((ftype, tuple(sorted(rolled, key=lambda lp: lp[0]))))
    return kept


def map_coedge_into_edges(pairings, indices):
    """Replace every co-edge index by its partner edge (pairings keys are strings in the JSON)."""
    return [pairings[str(i)] if str(i) in pairings else i for i in indices]


def filter_faces_by_coedge(pairings, faces):
    """Drop a face that reuses an edge already claimed through its co-edge partner (the reference
    defines it, post_processing.py:23-39, but never calls it)."""
    kept, used = [], set()
    for face in faces:
        drop = False
        for index in (i for loop in face[1] for i in loop):
            if index in pairings:
                index = pairings[index]
                if index in used:
                    drop = True
                    break
            used.add(index)
        if not drop:
            kept.append(face)
    return kept


def postprocess_faces(faces, edges, pairings, tol):
    """The is_coedge branch of the harness (trainer.py:226-255): enclosure filter, then flatten the
    loops and map co-edges onto edges."""
    closed = filter_faces_by_encloseness(edges, faces, tol)
    return [(ftype, map_coedge_into_edges(pairings, [i for loop in loops for i in loop]))
            for ftype, loops in closed]


# ---- JSON record -------------------------------------------------------------------------------------
def _plain(obj):
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.floating,)):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, (list, tuple)):
        return [_plain(x) for x in obj]
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    return obj


def faces_record(edges, dominant_directions, pred_faces, label_faces):
    """The per-sample dict the reference dumps for `reconstruction/` (trainer.py:126-133)."""
    return {"edges": _plain(edges), "dominant_directions": _plain(dominant_directions),
            "pred_faces": _plain(pred_faces), "label_faces": _plain(label_faces)}


def dumps_record(record):
    return json.dumps(_plain(record))
