"""CPU: wireframe JSON -> padded batch against items produced by the reference's
dataset classes (oracle/make_golden_data.py)."""
import json
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, token_ns
from faceformer_amd import datasets as D


def _cfg(c):
    return types.SimpleNamespace(token=token_ns(), **c)


def test_items_match_reference():
    meta = json.load(open(os.path.join(GOLDEN, "data_cases.json")))
    arrs = np.load(os.path.join(GOLDEN, "data_cases.npz"))
    cfg = _cfg(meta["cfg"])
    assert len(meta["cases"]) == 8
    for c in meta["cases"]:
        pack = D.pack_parallel_item if c["kind"] == "par" else D.pack_seq2seq_item
        item = pack(c["raw"], cfg)
        assert item["num_input"] == c["num_input"]
        for k in ("input", "label", "input_mask", "label_mask"):
            want = arrs[c["tag"] + "/" + k]
            assert item[k].shape == want.shape, (c["tag"], k)
            assert np.array_equal(item[k], want), (c["tag"], k)
        assert item["input"].dtype == np.float32 and item["input_mask"].dtype == bool


def test_dataset_classes_and_collate(tmp_path):
    meta = json.load(open(os.path.join(GOLDEN, "data_cases.json")))
    cfg = _cfg(meta["cfg"])
    names = []
    for i, c in enumerate([c for c in meta["cases"] if c["kind"] == "par"]):
        fn = "w%d.json" % i
        json.dump(c["raw"], open(tmp_path / fn, "w"))
        names.append(fn)
    open(tmp_path / "test.txt", "w").write("\n".join(names) + "\n")
    ds = D.ABCDataset_Parallel(str(tmp_path), ["test.txt"], cfg)
    assert len(ds) == 4 and ds[2]["name"] == "w2.json"
    batch = D.collate([ds[i] for i in range(4)])
    assert tuple(batch["input"].shape) == (4, 24, 50, 2) and batch["input_mask"].dtype == torch.bool
    assert isinstance(batch["num_input"], list) and tuple(batch["label"].shape) == (4, 24, 14)
    with pytest.raises(NotImplementedError):
        D.parse_splits_list(str(tmp_path), "a.csv")
    with pytest.raises(ValueError):
        D.pack_parallel_item({"edges": [[[0, 0], [1, 1]]] * 30, "faces_indices": []}, cfg)


def test_sample_points_shapes():
    assert D.sample_points([[0, 0], [1, 2]], 5).tolist() == [[0, 0], [0.25, 0.5], [0.5, 1.0], [0.75, 1.5], [1, 2]]
    poly = [[i, -i] for i in range(10)]
    assert D.sample_points(poly, 4).tolist() == [[0, 0], [3, -3], [6, -6], [9, -9]]
