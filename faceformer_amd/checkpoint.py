"""Weights loader for the reference's PyTorch-Lightning checkpoints.

A reference `.ckpt` is a torch pickle with `state_dict` (every key prefixed `model.` because the
harness stores the network as `self.model`, reference trainer.py:20) and `hyper_parameters` = the
pickled fvcore/yacs `CfgNode` (trainer.py:19).  Neither fvcore nor yacs nor pytorch_lightning is
installed on the target image, so unpickling maps those classes onto this package's `CfgNode` and
turns every other unknown class into an inert placeholder instead of importing it.
"""
import pickle
import types

import torch

from .config import CfgNode

__all__ = ["load_lightning_checkpoint", "model_from_checkpoint", "strip_prefix"]

_CFG_CLASSES = {("fvcore.common.config", "CfgNode"), ("yacs.config", "CfgNode"),
                ("detectron2.config.config", "CfgNode"),
                # checkpoints written with THIS package's own node (or through the `faceformer` alias package)
                ("faceformer_amd.config", "CfgNode"), ("faceformer.config", "CfgNode")}


class _Inert(dict):
    """Stand-in for classes of packages that are not installed (callbacks, loggers, ...)."""

    def __init__(self, *a, **k):
        super().__init__()

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state)


_BUILTINS_OK = {"dict", "list", "set", "frozenset", "tuple", "int", "float", "complex", "str", "bytes", "bytearray",
                "bool", "slice", "range", "object"}
_GLOBALS_OK = {("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "deque"),
               ("_codecs", "encode"), ("argparse", "Namespace"), ("copyreg", "_reconstructor")}


# numpy: exactly what an ndarray / numpy scalar pickle needs (both module spellings of numpy 1.x / 2.x) -- not the
# rest of numpy.core.multiarray (fromfile, frombuffer, copyto, ... are callables a crafted pickle could drive)
_NUMPY_OK = {(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")} | \
            {("numpy", "ndarray"), ("numpy", "dtype")}


def _torch_global_ok(obj):
    """Only what torch's own tensor serialisation needs: storage / tensor / parameter classes, dtype, size and
    device objects, and the `_rebuild_*` helpers -- never an arbitrary callable from the torch namespace."""
    if isinstance(obj, (torch.dtype,)):
        return True
    if isinstance(obj, type):
        ok = (torch.Tensor, torch.Size, torch.device, torch.dtype, torch.storage.TypedStorage,
              torch.storage.UntypedStorage)
        return issubclass(obj, ok) or obj.__name__.endswith("Storage")
    return callable(obj) and getattr(obj, "__name__", "").startswith("_rebuild")


class _Unpickler(pickle.Unpickler):
    """Allow-list unpickler: a downloaded .ckpt is a pickle and must not be able to import and call arbitrary
    code.  Config nodes map onto the local CfgNode; tensors / containers / numpy scalars are rebuilt; every
    other global (Lightning callbacks, loggers, optimiser classes, ...) becomes an inert placeholder."""

    def find_class(self, module, name):
        if (module, name) in _CFG_CLASSES:
            return CfgNode
        if module == "builtins":
            return super().find_class(module, name) if name in _BUILTINS_OK else _Inert
        if (module, name) in _GLOBALS_OK:
            return super().find_class(module, name)
        top = module.split(".")[0]
        if top == "torch":
            try:
                obj = super().find_class(module, name)
            except (ImportError, AttributeError):
                return _Inert
            return obj if _torch_global_ok(obj) else _Inert
        if (module, name) in _NUMPY_OK:
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return _Inert
        return _Inert


_pickle_module = types.ModuleType("faceformer_amd._ckpt_pickle")
_pickle_module.Unpickler = _Unpickler
_pickle_module.load = lambda f, **kw: _Unpickler(f, **kw).load()
_pickle_module.loads = pickle.loads
_pickle_module.dump, _pickle_module.dumps = pickle.dump, pickle.dumps
_pickle_module.__name__ = "pickle"


def strip_prefix(state_dict, prefix="model."):
    """Drop the Lightning module prefix; keys without it are kept unchanged."""
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def load_lightning_checkpoint(path, map_location="cpu"):
    """-> (state_dict without the `model.` prefix, hyper_parameters as CfgNode/dict or None)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_pickle_module)
    if "state_dict" not in ckpt:
        raise ValueError("%s is not a Lightning checkpoint (no 'state_dict')" % path)
    return strip_prefix(ckpt["state_dict"]), ckpt.get("hyper_parameters")


def model_from_checkpoint(path, cfg=None, model_class=None, device=None):
    """Build SurfaceFormer / SurfaceFormer_Parallel from a checkpoint.  `cfg` (a CfgNode with .model
    and .model_class) defaults to the checkpoint's own hyper-parameters."""
    from . import models
    sd, hp = load_lightning_checkpoint(path)
    cfg = cfg if cfg is not None else hp
    if cfg is None:
        raise ValueError("checkpoint carries no hyper_parameters; pass cfg")
    cls = model_class or getattr(models, cfg["model_class"])
    model = cls(**cfg["model"])
    model.load_state_dict(sd)
    model.eval()
    return model.to(device) if device is not None else model
