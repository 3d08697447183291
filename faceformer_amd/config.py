"""Config-file API of the decode path (mirror of the reference's `faceformer/config.py`).

The reference builds its defaults on fvcore's `CfgNode` (reference `faceformer/config.py:3-52`),
overlays a YAML file and a trailing `KEY VALUE ...` list, then freezes the tree
(`faceformer/config.py:73-79`).  fvcore/yacs are not available on the target image, so this module
carries a small self-contained node type with the same observable behaviour for this path:

* attribute access and mapping access (`cfg.model.num_lines`, `cfg["model"]`, `**cfg.model`);
* `clone()`, `merge_from_file(path)`, `merge_from_list([k, v, ...])`, `freeze()`, `defrost()`;
* merging refuses keys that do not exist in the defaults (yacs raises `KeyError` as well) and
  type-checks overriding values the way yacs does (int -> float promotion allowed, list <-> tuple).

`get_parser()` / `get_cfg(args)` keep the reference's flags and semantics
(`faceformer/config.py:54-79`).
"""
import argparse
import ast
import copy

import yaml

__all__ = ["CfgNode", "CN", "get_parser", "get_cfg", "default_cfg"]


class CfgNode(dict):
    """Attribute-style nested config dictionary with freeze / merge support."""

    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # -- attribute protocol ------------------------------------------------------------------
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.is_frozen():
            raise AttributeError(
                "Attempted to set {} to {}, but CfgNode is immutable".format(name, value))
        self[name] = value

    def __setitem__(self, key, value):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable (frozen); cannot set '{}'".format(key))
        super().__setitem__(key, value)

    # pickling / deepcopy must not go through __getattr__
    def __getstate__(self):
        return {"items": dict(self), "frozen": self.is_frozen()}

    def __setstate__(self, state):
        """Own pickles carry {'items', 'frozen'}.  A FOREIGN node (fvcore / yacs `CfgNode`, which
        checkpoint.py maps onto this class) arrives through the dict-subclass protocol instead: its
        entries were already stored by SETITEMS and `state` is the instance __dict__ of yacs
        ({'__immutable__', '__deprecated_keys__', '__renamed_keys__', '__new_allowed__'}); only the
        immutable flag means anything here."""
        object.__setattr__(self, CfgNode._FROZEN, False)
        if isinstance(state, dict) and "items" in state and "frozen" in state:
            for k, v in state["items"].items():
                dict.__setitem__(self, k, v)
            object.__setattr__(self, CfgNode._FROZEN, bool(state["frozen"]))
        elif isinstance(state, dict):
            object.__setattr__(self, CfgNode._FROZEN, bool(state.get("__immutable__", False)))

    def __reduce__(self):
        return (CfgNode, (), self.__getstate__())

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, CfgNode._FROZEN, self.is_frozen())
        return out

    # -- freeze ------------------------------------------------------------------------------
    def is_frozen(self):
        # instances revived by a foreign pickle (dict-subclass protocol: __new__ + SETITEMS, no
        # __init__) have no flag yet: they count as mutable
        return self.__dict__.get(CfgNode._FROZEN, False)

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def clone(self):
        out = copy.deepcopy(self)
        return out

    # -- merging -----------------------------------------------------------------------------
    @staticmethod
    def _coerce(new, old, key):
        """yacs-style value check: the overriding value must match the default's type."""
        if old is None or new is None:
            return new
        told, tnew = type(old), type(new)
        if told is tnew:
            return new
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, CfgNode) and isinstance(new, dict):
            return CfgNode(new)
        raise ValueError(
            "Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}".format(
                told, tnew, old, new, key))

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: {}".format(full))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("Config key {} expects a mapping".format(full))
                self[k]._merge(v, path + [k])
            else:
                self[k] = CfgNode._coerce(v, self[k], full)

    def merge_from_other_cfg(self, other):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable (frozen)")
        self._merge(other, [])

    def merge_from_file(self, path):
        with open(path, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self.merge_from_other_cfg(loaded)

    @staticmethod
    def _decode(value):
        """Turn a command-line string into a python literal when it parses as one."""
        if not isinstance(value, str):
            return value
        try:
            return ast.literal_eval(value)
        except (ValueError, SyntaxError):
            return value

    def merge_from_list(self, opts):
        if self.is_frozen():
            raise AttributeError("CfgNode is immutable (frozen)")
        opts = list(opts or [])
        if len(opts) % 2 != 0:
            raise ValueError(
                "Override list has odd length: {}; it must be a list of pairs".format(opts))
        for full, raw in zip(opts[0::2], opts[1::2]):
            node = self
            parts = full.split(".")
            for p in parts[:-1]:
                if p not in node or not isinstance(node[p], CfgNode):
                    raise KeyError("Non-existent config key: {}".format(full))
                node = node[p]
            leaf = parts[-1]
            if leaf not in node:
                raise KeyError("Non-existent config key: {}".format(full))
            node[leaf] = CfgNode._coerce(CfgNode._decode(raw), node[leaf], full)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, CfgNode) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self), sort_keys=True)

    def __repr__(self):
        return "CfgNode({})".format(dict.__repr__(self))


CN = CfgNode


def default_cfg():
    """Default tree; key names and values follow reference `faceformer/config.py:7-52`."""
    return CN({
        "model_class": "SurfaceFormer",
        "dataset_class": "ABCDataset",
        "root_dir": "/root/data",
        "batch_size_train": 64,
        "batch_size_valid": 128,
        "datasets_train": ["train.txt"],
        "datasets_valid": ["valid.txt"],
        "datasets_test": ["test.txt"],
        "trainer": {
            "name": "surfaceformer",
            "version": "baseline",
            "num_gpus": [0],
            "precision": 16,
            "checkpoint_period": 2,
            "lr": 1e-3,
            "lr_step": 0,
        },
        "model": {
            "num_points_per_line": 50,
            "num_lines": 64,
            "point_dim": 2,
            "label_seq_length": 128,
            "max_num_faces": 42,
            "max_face_length": 34,
            "num_model": 512,
            "num_head": 8,
            "num_feedforward": 1024,
            "num_encoder_layers": 6,
            "num_decoder_layers": 6,
            "dropout": 0.2,
            "token": {
                "PAD": 0,
                "SOS": 1,
                "SEP": 2,
                "EOS": 3,
                "DIR0": 4,
                "DIR1": 5,
                "len": 4,
                "face_type_offset": 1,
            },
        },
        "post_process": {
            "enclosedness_tol": 2e-4,
            "is_coedge": True,
        },
    })


_C = default_cfg()


def get_parser():
    """Same flags as reference `faceformer/config.py:54-70`."""
    parser = argparse.ArgumentParser(description="SurfaceFormer decode (MI355X-native)")
    parser.add_argument("--config-file", default="", metavar="FILE", help="path to config file")
    parser.add_argument("--valid_ckpt", default="", help="path to validation checkpoint")
    parser.add_argument("--test_ckpt", default="", help="path to testing checkpoint")
    parser.add_argument("--resume_ckpt", default="",
                        help="path to training checkpoint (training is out of scope here)")
    parser.add_argument("opts", default=None, nargs=argparse.REMAINDER,
                        help="Modify config options using the command-line")
    return parser


def get_cfg(args):
    """defaults -> YAML overlay -> opts overlay -> frozen (reference `config.py:73-79`)."""
    cfg = _C.clone()
    cfg.defrost()
    if getattr(args, "config_file", ""):
        cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(getattr(args, "opts", None))
    cfg.freeze()
    return cfg


def load_cfg(config_file="", opts=None):
    """Convenience wrapper: `load_cfg('configs/ours.yml', ['model.num_lines', 256])`."""
    return get_cfg(argparse.Namespace(config_file=config_file, opts=list(opts or [])))
