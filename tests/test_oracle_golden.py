"""CPU: the oracle (oracle/refpath.py) must reproduce the golden vectors that were captured from the
imported reference (oracle/make_golden.py) bit for bit -- predict, per-step best logit / margin and
the stored masked logits."""
import numpy as np
import pytest
import torch

from conftest import case_weights_and_batch, ctor_kwargs, golden_names, load_golden
from oracle import refpath


def _check(name):
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    trace = {}
    extra = batch.pop("extra_mask", None)
    if case["kind"] == "parallel":
        out = refpath.parallel_forward_eval(sd, batch, num_head=case["model"]["H"], trace=trace, extra_mask=extra,
                                            **ctor_kwargs(case))
    else:
        out = refpath.seq2seq_forward_eval(sd, batch, num_head=case["model"]["H"], trace=trace, extra_mask=extra,
                                           **ctor_kwargs(case))
    assert np.array_equal(out["predict"].numpy(), z["predict"])
    assert len(trace["logits"]) == int(z["steps"])
    rows = z["logit_rows"]
    got = torch.stack([l[rows] for l in trace["logits"]]).numpy()
    assert np.array_equal(got, z["logits"])
    best = torch.stack([l.max(dim=1).values for l in trace["logits"]]).numpy()
    assert np.array_equal(best, z["best"])
    mem = trace["memory"]
    if "memory" in z:
        assert np.array_equal(mem.numpy(), z["memory"])
    else:
        assert np.array_equal(mem[:, :8].numpy(), z["memory_head"])
    if case["kind"] == "seq2seq":
        assert np.array_equal(out["pointer"][:, -1].numpy(), z["pointer_last"])


@pytest.mark.parametrize("name", golden_names(include_slow=False) + golden_names(include_slow=False, module_path=True))
def test_oracle_matches_golden(name):
    _check(name)


@pytest.mark.slow
def test_oracle_matches_golden_seq2seq_config_a():
    _check("seq_full_A64_gain4")


@pytest.mark.slow
def test_oracle_matches_golden_seq2seq_config_d_extra_mask():
    _check("seq_full_D216_extramask")


def test_anchor_limit_is_a_faithful_sample():
    """cpu_baseline times a subset of anchor sequences; their tokens must equal the full run's."""
    case, z = load_golden("par_small_gain4")
    sd, batch = case_weights_and_batch(case)
    out = refpath.parallel_forward_eval(sd, batch, num_head=case["model"]["H"], anchor_limit=5)
    full = z["predict"]
    steps = int(z["steps"])
    assert np.array_equal(out["predict"].numpy()[:, :, : steps + 1], full[:, :5, : steps + 1])
