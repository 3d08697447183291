"""Input side of the path: wireframe JSON -> padded tensors with the reference's
batch keys.  Mirrors `ABCDataset` / `ABCDataset_Parallel` (reference faceformer/datasets/data.py:31-97,
data_para.py:28-110): every edge polyline is resampled to `num_points_per_line` points (two-point
edges by linear interpolation, longer ones by index rounding), zero-padded to `num_lines`; labels are
built from `faces_indices` (token offset, SEP/EOS or face-type terminators, loop rotations).  Host-side
numpy, like the reference; the result feeds `model(batch)` after `.cuda()`.
"""
import json
import os

import numpy as np
import torch

__all__ = ["sample_points", "pack_seq2seq_item", "pack_parallel_item", "collate", "ABCDataset",
           "ABCDataset_Parallel", "parse_splits_list"]


def sample_points(edge, num_samples=50):
    """2-point edge: `num_samples` evenly spaced points on the segment; polyline: pick
    round(linspace(0, len-1)) of its points (reference data.py:11-28)."""
    if len(edge) == 2:
        t = np.linspace(0, 1, num_samples)
        (x1, y1), (x2, y2) = edge[0][:2], edge[1][:2]
        return np.vstack([x1 + (x2 - x1) * t, y1 + (y2 - y1) * t]).T
    pick = np.linspace(0, len(edge) - 1, num_samples).round(0).astype(int)
    return np.array(edge)[pick]


def _pack_input(edges, cfg):
    P, L, D = cfg.num_points_per_line, cfg.num_lines, cfg.point_dim
    if len(edges) > L:
        raise ValueError("wireframe with %d edges exceeds num_lines=%d" % (len(edges), L))
    inp = np.zeros((L, P, D), dtype=np.float32)
    for i, edge in enumerate(edges):
        inp[i, :P] = sample_points(edge, P)
    mask = np.ones(L, dtype=bool)
    mask[: len(edges)] = False
    return inp, mask


def _flatten(nested):
    return [x for sub in nested for x in sub]


def pack_seq2seq_item(raw, cfg, index=0, name=""):
    """One sample of `ABCDataset`: label = SOS f1 SEP f2 SEP ... EOS, indices shifted by token.len."""
    tok = cfg.token
    edges, faces = raw["edges"], raw["faces_indices"]
    inp, mask = _pack_input(edges, cfg)
    label = np.full(cfg.label_seq_length, tok.PAD, dtype=np.int64)
    label[0] = tok.SOS
    pos = 0
    for face in faces:
        if not isinstance(face[0], int):
            face = _flatten(face)
        pos += 1
        label[pos: pos + len(face)] = np.asarray(face, dtype=np.int64) + tok.len
        pos += len(face)
        label[pos] = tok.SEP
    label[pos] = tok.EOS
    return {"id": index, "input": inp, "label": label, "num_input": len(edges), "num_label": pos + 1,
            "input_mask": mask, "label_mask": label == tok.PAD, "name": name}


def pack_parallel_item(raw, cfg, index=0, name=""):
    """One sample of `ABCDataset_Parallel`: one label row per (loop, rotation) of every face: the
    rotated loop, then the face's other loops, then the face-type token (types > 1 collapse to 2,
    plus face_type_offset); unused rows start with token.len-1."""
    tok = cfg.token
    edges, faces = raw["edges"], raw["faces_indices"]
    inp, mask = _pack_input(edges, cfg)
    L, T = cfg.num_lines, cfg.max_face_length
    label = np.full((L, T), tok.PAD, dtype=np.int64)
    row = 0
    for ftype, face in faces:
        ftype = (2 if ftype > 1 else ftype) + tok.face_type_offset
        for loop in face:
            for shift in range(len(loop)):
                seq = np.roll(loop, shift, axis=0).tolist()
                for other in face:
                    if other != loop:
                        seq += other
                label[row, : len(seq)] = np.asarray(seq, dtype=np.int64) + tok.len
                label[row, len(seq)] = ftype
                row += 1
    label[row:, 0] = tok.len - 1
    return {"id": index, "input": inp, "label": label, "num_input": len(edges), "num_faces": len(faces),
            "input_mask": mask, "label_mask": label == tok.PAD, "name": name}


def collate(items):
    """Default-collate equivalent for the keys the path reads."""
    out = {}
    for k in items[0]:
        vals = [it[k] for it in items]
        if isinstance(vals[0], np.ndarray):
            out[k] = torch.from_numpy(np.stack(vals))
        elif isinstance(vals[0], (int, np.integer)):
            out[k] = [int(v) for v in vals] if k == "num_input" else torch.tensor(vals)
        else:
            out[k] = vals
    return out


def parse_splits_list(root_dir, splits):
    """'.json' entries are used as they are, '.txt' entries list json paths relative to root_dir
    (reference data.py:99-118)."""
    if isinstance(splits, str):
        splits = splits.split()
    files = []
    for split in splits:
        ext = os.path.splitext(split)[1]
        path = os.path.join(root_dir, split)
        if ext == ".json":
            files.append(path)
        elif ext == ".txt":
            with open(path) as f:
                files += [line.rstrip() for line in f]
        else:
            raise NotImplementedError("%s not a valid info_file type" % split)
    return files


class _ABCBase(torch.utils.data.Dataset):
    pack = None

    def __init__(self, root_dir, datafile_path, config):
        super().__init__()
        self.root_dir, self.config = root_dir, config
        self.info_files = parse_splits_list(root_dir, datafile_path)
        self.token = config.token
        self.raw_datas = []
        for info in self.info_files:
            with open(os.path.join(root_dir, info)) as f:
                self.raw_datas.append(json.load(f))

    def __len__(self):
        return len(self.info_files)

    def __getitem__(self, index):
        return type(self).pack(self.raw_datas[index], self.config, index, self.info_files[index])


class ABCDataset(_ABCBase):
    pack = staticmethod(pack_seq2seq_item)


class ABCDataset_Parallel(_ABCBase):
    pack = staticmethod(pack_parallel_item)
