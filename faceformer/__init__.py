"""Alias package: `faceformer.models`, `faceformer.transformer`, `faceformer.embedding`,
`faceformer.utils`, `faceformer.config` resolve to the MI355X-native implementations in
`faceformer_amd`, so code written against the reference's import paths runs unchanged."""
import importlib
import sys

import faceformer_amd as _impl

for _name in ("config", "utils", "embedding", "transformer", "models", "datasets", "post_processing"):
    _mod = importlib.import_module("faceformer_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    setattr(sys.modules[__name__], _name, _mod)
sys.modules[__name__ + ".models.model"] = importlib.import_module("faceformer_amd.models.model")
sys.modules[__name__ + ".models.model_para"] = importlib.import_module("faceformer_amd.models.model_para")

__all__ = ["config", "utils", "embedding", "transformer", "models", "datasets", "post_processing"]
