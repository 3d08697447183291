"""CPU: Lightning-style checkpoint loader on a synthetic checkpoint whose
hyper-parameters are pickled as `fvcore.common.config.CfgNode` -- a package that is NOT installed."""
import sys
import types

import pytest
import torch

from conftest import token_ns


def _write_fake_ckpt(path, model, cfg_dict):
    mod_a, mod_b, mod_c = types.ModuleType("fvcore"), types.ModuleType("fvcore.common"), types.ModuleType("fvcore.common.config")

    class CfgNode(dict):
        """Shaped like yacs.config.CfgNode: a dict subclass whose instance __dict__ holds the bookkeeping
        keys, so pickling emits a BUILD with that state after the items (what real reference .ckpt files do)."""

        def __init__(self, init=None):
            super().__init__(init or {})
            self.__dict__["__immutable__"] = False
            self.__dict__["__deprecated_keys__"] = set()
            self.__dict__["__renamed_keys__"] = {}
            self.__dict__["__new_allowed__"] = False

        def freeze(self):
            self.__dict__["__immutable__"] = True
            for v in self.values():
                if isinstance(v, CfgNode):
                    v.freeze()
    CfgNode.__module__, CfgNode.__qualname__ = "fvcore.common.config", "CfgNode"
    mod_c.CfgNode = CfgNode
    pl_mod = types.ModuleType("pytorch_lightning_fake_callbacks")

    class ModelCheckpoint:
        def __init__(self):
            self.best = 0.5
    ModelCheckpoint.__module__, ModelCheckpoint.__qualname__ = "pytorch_lightning_fake_callbacks", "ModelCheckpoint"
    pl_mod.ModelCheckpoint = ModelCheckpoint
    sys.modules.update({"fvcore": mod_a, "fvcore.common": mod_b, "fvcore.common.config": mod_c,
                        "pytorch_lightning_fake_callbacks": pl_mod})
    try:
        def to_node(d):
            return CfgNode({k: to_node(v) if isinstance(v, dict) else v for k, v in d.items()})
        hp = to_node(cfg_dict)
        hp.freeze()                     # the reference freezes its config before Lightning pickles it
        assert "__immutable__" in hp.__dict__ and isinstance(hp["model"], CfgNode)
        ckpt = {"epoch": 3, "state_dict": {"model." + k: v for k, v in model.state_dict().items()},
                "hyper_parameters": hp, "callbacks": {"ckpt": ModelCheckpoint()}}
        torch.save(ckpt, path)
    finally:
        for k in ("fvcore", "fvcore.common", "fvcore.common.config", "pytorch_lightning_fake_callbacks"):
            sys.modules.pop(k, None)


def test_load_checkpoint_without_fvcore(tmp_path):
    from faceformer_amd.checkpoint import load_lightning_checkpoint, model_from_checkpoint
    from faceformer_amd.config import CfgNode
    from faceformer_amd.models import SurfaceFormer_Parallel
    tok = dict(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    mcfg = dict(num_model=128, num_head=2, num_feedforward=256, num_encoder_layers=1, num_decoder_layers=2,
                dropout=0.2, num_points_per_line=50, num_lines=12, point_dim=2, max_face_length=6,
                max_num_faces=42, label_seq_length=20, token=tok)
    model = SurfaceFormer_Parallel(**{**mcfg, "token": token_ns()})
    path = str(tmp_path / "last.ckpt")
    _write_fake_ckpt(path, model, {"model_class": "SurfaceFormer_Parallel", "root_dir": "/stale/path", "model": mcfg})
    assert "fvcore" not in sys.modules
    sd, hp = load_lightning_checkpoint(path)
    assert isinstance(hp, CfgNode) and hp.model.token.len == 4 and hp["root_dir"] == "/stale/path"
    assert isinstance(hp.model, CfgNode) and isinstance(hp.model.token, CfgNode)
    assert hp.is_frozen() and hp.model.is_frozen()          # yacs' __immutable__ survives as the frozen flag
    with pytest.raises(AttributeError):
        hp.model.num_lines = 3
    assert set(sd) == set(model.state_dict()) and not any(k.startswith("model.") for k in sd)
    m2 = model_from_checkpoint(path)
    assert isinstance(m2, SurfaceFormer_Parallel) and not m2.training
    for k, v in model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    torch.save({"foo": 1}, str(tmp_path / "bad.ckpt"))
    with pytest.raises(ValueError):
        load_lightning_checkpoint(str(tmp_path / "bad.ckpt"))


def test_checkpoint_unpickler_refuses_arbitrary_globals(tmp_path):
    """A .ckpt is a pickle: a global outside the allow-list must come back as an inert placeholder, not be
    imported and called."""
    import os
    from faceformer_amd.checkpoint import load_lightning_checkpoint

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))
    path = str(tmp_path / "evil.ckpt")
    torch.save({"state_dict": {"model.w": torch.ones(2)}, "hyper_parameters": {"x": 1}, "payload": Evil()}, path)
    sd, hp = load_lightning_checkpoint(path)
    assert not (tmp_path / "pwned").exists()
    assert torch.equal(sd["w"], torch.ones(2)) and hp["x"] == 1


def test_checkpoint_with_this_packages_own_cfgnode_round_trips(tmp_path):
    """hyper_parameters pickled as faceformer_amd.config.CfgNode (a checkpoint written with THIS package, also
    through the `faceformer` alias) must come back as a CfgNode, not as an inert placeholder."""
    from faceformer_amd.checkpoint import load_lightning_checkpoint, model_from_checkpoint
    from faceformer_amd.config import CfgNode
    from faceformer_amd.models import SurfaceFormer
    tok = dict(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)
    mcfg = dict(num_model=128, num_head=2, num_feedforward=256, num_encoder_layers=1, num_decoder_layers=1,
                num_points_per_line=50, num_lines=10, point_dim=2, label_seq_length=9, token=tok)
    model = SurfaceFormer(**{**mcfg, "token": token_ns()})
    hp = CfgNode({"model_class": "SurfaceFormer", "model": mcfg})
    hp.freeze()
    path = str(tmp_path / "own.ckpt")
    torch.save({"state_dict": {"model." + k: v for k, v in model.state_dict().items()}, "hyper_parameters": hp}, path)
    sd, hp2 = load_lightning_checkpoint(path)
    assert type(hp2) is CfgNode and hp2.is_frozen() and hp2.model.token.EOS == 3 and hp2["model_class"] == "SurfaceFormer"
    m2 = model_from_checkpoint(path)
    assert isinstance(m2, SurfaceFormer)
    for k, v in model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_checkpoint_numpy_allow_list_is_narrow(tmp_path):
    """ndarrays / numpy scalars inside a checkpoint are rebuilt; any other numpy callable (fromfile, frombuffer,
    ...) is NOT imported."""
    import numpy as np
    from faceformer_amd.checkpoint import _Inert, _Unpickler, load_lightning_checkpoint
    path = str(tmp_path / "np.ckpt")
    torch.save({"state_dict": {"model.w": torch.ones(2)}, "hyper_parameters": {"x": 1},
                "arr": np.arange(6, dtype=np.float32).reshape(2, 3), "scalar": np.float64(2.5)}, path)
    import pickle, zipfile
    with zipfile.ZipFile(path) as z:
        name = [n for n in z.namelist() if n.endswith("data.pkl")][0]
        raw = z.read(name)
    assert b"numpy" in raw
    ck = torch.load(path, map_location="cpu", weights_only=False,
                    pickle_module=__import__("faceformer_amd.checkpoint", fromlist=["_pickle_module"])._pickle_module)
    assert isinstance(ck["arr"], np.ndarray) and ck["arr"].shape == (2, 3) and float(ck["scalar"]) == 2.5
    import io
    up = _Unpickler(io.BytesIO(b""))
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray", "numpy"):
        for bad in ("fromfile", "frombuffer", "copyto", "fromstring", "load"):
            assert up.find_class(mod, bad) is _Inert, (mod, bad)
    assert up.find_class("numpy", "ndarray") is np.ndarray
    sd, _ = load_lightning_checkpoint(path)
    assert torch.equal(sd["w"], torch.ones(2))
