import json
import os
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU oracle case")


def has_gpu():
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need a ROCm device: on a host without one a plain `pytest tests` skips them (with the
    reason) instead of failing in torch._C._cuda_init; `-m gpu` on a GPU box runs them all."""
    if has_gpu():
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built on demand: hipcc cross-compiles without a GPU)."""
    from faceformer_amd.hip import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build()
    return lib.load()


def token_ns():
    return types.SimpleNamespace(PAD=0, SOS=1, SEP=2, EOS=3, DIR0=4, DIR1=5, len=4, face_type_offset=1)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = json.loads(bytes(z["case"]).decode())
    return case, z


def is_module_path(case):
    """Constructor arguments the native engine does not implement (post-norm layers, gelu, head widths other than 64):
    `forward_eval` then runs the reference's loop on the HIP sub-modules (models/common.py: _forward_eval_modules)."""
    m = case["model"]
    return (not m.get("normalize_before", True)) or m.get("activation", "relu") != "relu" or m["E"] != 64 * m["H"]


def golden_names(include_slow=True, module_path=False):
    """Names of the golden cases: the ones the native engine decodes (default) or the ones that take the sub-module loop."""
    from oracle.golden_cases import CASES
    return [c["name"] for c in CASES if (include_slow or not c.get("slow")) and is_module_path(c) == module_path]


def ctor_kwargs(case):
    m = case["model"]
    return dict(normalize_before=m.get("normalize_before", True), activation=m.get("activation", "relu"))


def case_weights_and_batch(case):
    """Regenerate (state_dict, batch) of a golden case from its seeds."""
    from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec
    m = case["model"]
    spec = state_dict_spec(case["kind"], m["L"], m["seq_len"], m["E"], m["FF"], m["enc"], m["dec"],
                           encoder_norm=m.get("normalize_before", True))
    sd = make_state_dict(spec, case["recipe"], case["wseed"])
    batch = make_wireframes(case["n_edges"], m["L"], m["seq_len"], case["kind"], seeds=case["seeds"])
    if case.get("extra_mask_seed") is not None:
        from faceformer_amd.synth import make_extra_mask
        batch["extra_mask"] = make_extra_mask(case, batch)
    return sd, batch


def build_model(case, sd, device):
    from faceformer_amd.models import SurfaceFormer, SurfaceFormer_Parallel
    m = case["model"]
    common = dict(num_model=m["E"], num_head=m["H"], num_feedforward=m["FF"],
                  num_encoder_layers=m["enc"], num_decoder_layers=m["dec"], dropout=0.2,
                  num_lines=m["L"], token=token_ns(), **ctor_kwargs(case))
    if case["kind"] == "parallel":
        model = SurfaceFormer_Parallel(max_face_length=m["seq_len"], **common)
    else:
        model = SurfaceFormer(label_seq_length=m["seq_len"], **common)
    model.load_state_dict(sd)
    return model.eval().to(device)


def batch_to(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
