"""CPU, world_size 2 over gloo: the multi-GPU result path (shard by wireframe, global stop rule from
all-reduced counters, all-gather of tokens) must reproduce the single-process tensor.  The device
engine is replaced by an oracle-backed stand-in (tests only) because no GPU exists here; the GPU
variant of this test lives in test_parity_golden.py::test_sharded_equals_single."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, case_weights_and_batch, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _OracleEngine:
    """Stand-in for PathEngine: same decode() contract, tokens from the CPU oracle."""

    def __init__(self, sd, num_head, batch):
        self.sd, self.H, self.batch = sd, num_head, batch

    def decode(self, memory, mask, kv_len, variant, T, F, num_input, no_stop=False, stop_callback=None, sync_every=0, **kw):
        """ff_decode's contract without the local stop rule: either every step (no_stop), or the steps enqueued when the
        caller's rule -- asked every sync_every steps, one period behind (faceformer_amd.dist.check_points) -- says stop."""
        from oracle import refpath
        from faceformer_amd.dist import check_points
        assert no_stop or stop_callback is not None
        sub_in = memory          # the stand-in's "memory" is the sub-batch handed to _encode
        n = sub_in["input"].size(0)
        sub = {"input": sub_in["input"], "input_mask": sub_in["input_mask"],
               "label": self.batch["label"][:n], "num_input": list(num_input)}
        trace = {}
        out = refpath.parallel_forward_eval(self.sd, sub, num_head=self.H, trace=trace, stop_rule=False,
                                            num_anchors=F)
        pred, counts, executed = out["predict"].reshape(-1, T).clone(), list(trace["counts"]), T - 1
        if stop_callback is not None:
            for enq, counted in check_points(T, sync_every):
                if stop_callback(counts[:counted]):
                    executed = enq
                    break
            pred[:, executed + 1:] = 0
        self.executed = executed
        return {"predict": pred, "steps": executed, "step_counts": counts[:executed]}


def _worker(rank, world, port, name, ret, local=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from faceformer_amd import dist as ffd
    from faceformer_amd.models import SurfaceFormer_Parallel
    from conftest import token_ns
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    case, z = load_golden(name)
    sd, batch = case_weights_and_batch(case)
    m = case["model"]
    model = SurfaceFormer_Parallel(num_model=m["E"], num_head=m["H"], num_feedforward=m["FF"],
                                   num_encoder_layers=m["enc"], num_decoder_layers=m["dec"],
                                   num_lines=m["L"], max_face_length=m["seq_len"], token=token_ns()).eval()
    eng = _OracleEngine(sd, m["H"], batch)
    model._encode = lambda sub: (eng, sub, None, None)   # carries the rank's sub-batch to the stand-in
    model.sharded_sync_every = 1 if local != "nochecks" else 0    # (T - 1 is 7..8 here: ask the global rule at every step)
    if local is True or local == "mismatch":
        # every rank holds ONLY its own wireframes (rank 0 the first one, rank 1 the rest): F, the counters and
        # the shard sizes are agreed by collectives; the result is the concatenation in rank order
        N = batch["input"].size(0)
        lo, hi = (0, 1) if rank == 0 else (1, N)
        mine = {"input": batch["input"][lo:hi], "input_mask": batch["input_mask"][lo:hi],
                "label": batch["label"][lo:hi], "num_input": batch["num_input"][lo:hi]}
        if local == "mismatch":
            # rank 1 padded its wireframes to a narrower num_lines: must be refused on EVERY rank, before any decode
            if rank == 1:
                mine = dict(mine, input=mine["input"][:, :-2], input_mask=mine["input_mask"][:, :-2])
            try:
                ffd.decode_sharded(model, mine, dist, local_shard=True)
                ret[rank] = False
            except ValueError as e:
                ret[rank] = "different num_lines" in str(e)
            dist.destroy_process_group()
            return
        out = ffd.decode_sharded(model, mine, dist, local_shard=True)
        ret[rank] = bool(np.array_equal(out["predict"].numpy(), z["predict"]) and out["shard_sizes"] == [1, N - 1])
        dist.destroy_process_group()
        return
    out = ffd.decode_sharded(model, dict(batch), dist)
    ok = np.array_equal(out["predict"].numpy(), z["predict"])
    # the periodic global check ends the decode on EVERY rank at most two periods behind the reference's stop step
    steps = int(z["steps"])
    T = m["seq_len"]
    if hasattr(eng, "executed") and model.sharded_sync_every > 0:
        ok = ok and eng.executed <= min(T - 1, max(steps + 2, 2))
        if steps + 2 < T - 1:
            ok = ok and eng.executed < T - 1
    elif hasattr(eng, "executed"):
        ok = ok and eng.executed == T - 1
    # face-loop JSON of every wireframe on every rank (parsed locally, gathered as bytes)
    import json
    from faceformer_amd import faces as FZ
    recs = ffd.decode_to_face_json(model, dict(batch), dist)
    ok = ok and len(recs) == batch["input"].size(0)
    for i, r in enumerate(recs):
        pf, _ = FZ.parse_parallel_faces(z["predict"][i], batch["label"][i].numpy(), batch["num_input"][i], token_ns())
        want = [[t, list(f)] for t, f in FZ.unique_faces_with_majority_type(pf)]
        ok = ok and json.loads(r)["pred_faces"] == want
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["par_small_ragged", "par_small_earlybreak", "par_small_break1"])
def test_sharded_decode_equals_single_process_gloo(name):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("name,mode", [("par_small_break1", False), ("par_small_earlybreak", False), ("par_small_break1", "nochecks")])
def test_sharded_decode_with_an_idle_rank_gloo(name, mode):
    """Three ranks, two wireframes: rank 2 decodes nothing and must still meet the other ranks at every periodic stop check
    (zeros at the same check points) -- or, with the checks off, only at the final collectives."""
    world = 3
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, ret, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True, 2: True}


@pytest.mark.parametrize("name", ["par_small_ragged", "par_small_earlybreak"])
def test_sharded_decode_with_shard_local_inputs_gloo(name):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, ret, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def test_shard_local_inputs_of_different_padded_width_are_refused_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "par_small_ragged", ret, "mismatch")) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True}


def _json_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from faceformer_amd import dist as ffd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    mine = [["a" * 5, "{\"k\": [1, 2]}"], [], ["\u00e9" * 300]][rank]
    got = ffd.gather_json_records(mine, dist)
    ret[rank] = got == ["a" * 5, "{\"k\": [1, 2]}", "\u00e9" * 300]
    dist.destroy_process_group()


def test_gather_json_records_ragged_gloo():
    world = 3
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_json_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(ret) == {0: True, 1: True, 2: True}


def test_shard_plan_balances_ragged_batches_by_cost():
    from faceformer_amd import dist as ffd
    # uniform batch: contiguous blocks (config C: 1024 wireframes over 8 GPUs -> 128 each)
    plan = ffd.shard_plan([256] * 1024, 8)
    assert [len(p) for p in plan] == [128] * 8 and plan[3][0] == 384 and plan[3][-1] == 511
    # ragged batch (config E mix): every wireframe exactly once, per-rank cost within 15 % of the mean
    rng = np.random.default_rng(0)
    ni = rng.choice([64, 128, 256, 512, 1024], size=256, p=[.3, .3, .2, .1, .1]).tolist()
    plan = ffd.shard_plan(ni, 8, F=1024)
    assert sorted(i for p in plan for i in p) == list(range(256))
    cost = [sum(ffd.wireframe_cost(ni[i], 1024) for i in p) for p in plan]
    assert max(cost) <= 1.15 * (sum(cost) / 8)
    by_index = [sum(ffd.wireframe_cost(ni[i], 1024) for i in range(*ffd.shard_range(256, r, 8)[:2])) for r in range(8)]
    assert max(cost) <= max(by_index)          # never worse than splitting by index
    for p in plan:                              # each rank sees its wireframes widest first
        assert [ni[i] for i in p] == sorted((ni[i] for i in p), reverse=True)


def test_shard_range_and_global_stop():
    from faceformer_amd import dist as ffd
    from faceformer_amd.hip import lib as L
    assert [ffd.shard_range(5, r, 2)[:2] for r in range(2)] == [(0, 3), (3, 5)]
    assert [ffd.shard_range(3, r, 4)[:2] for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert ffd.shard_range(1024, 7, 8) == (896, 1024, 128)
    pred = torch.arange(1, 13).view(2, 6).clone()
    out, stop = ffd.apply_global_stop(pred.clone(), [3, 0, 2, 1, 9], 2, L.FF_PARALLEL)
    assert stop == 2 and out[:, 3:].eq(0).all() and out[:, :3].equal(pred[:, :3])
    out, stop = ffd.apply_global_stop(pred.clone(), [1, 0, 1, 0, 0], 2, L.FF_SEQ2SEQ)
    assert stop == 3 and out[:, 4:].eq(0).all()
    out, stop = ffd.apply_global_stop(pred.clone(), [1, 2, 0, 0, 0], 2, L.FF_SEQ2SEQ)
    assert stop == 5 and out.equal(pred)   # the count jumped past N: the reference never stops early


def test_check_points_and_stop_step():
    """The cadence every rank of a sharded decode replays (ff_engine.hip: rule looked at every k steps, one period behind the
    enqueued steps) and the reference's loop-break rule on batch-global counters."""
    from faceformer_amd import dist as ffd
    from faceformer_amd.hip import lib as L
    assert ffd.check_points(10, 1) == [(e, e - 1) for e in range(2, 9)]
    assert ffd.check_points(37, 4) == [(8, 4), (12, 8), (16, 12), (20, 16), (24, 20), (28, 24), (32, 28)]
    assert ffd.check_points(37, 0) == [] and ffd.check_points(3, 1) == [] and ffd.check_points(9, 4) == []
    assert ffd.stop_step([3, 1, 0, 5], 2, L.FF_PARALLEL) == 3 and ffd.stop_step([3, 1], 2, L.FF_PARALLEL) is None
    assert ffd.stop_step([0, 1, 0, 1], 2, L.FF_SEQ2SEQ) == 4 and ffd.stop_step([1, 2, 0], 2, L.FF_SEQ2SEQ) is None
    assert ffd.stop_step([], 2, L.FF_PARALLEL) is None
    # apply_global_stop and stop_step agree on where the loop breaks
    for counts, n, v in (([3, 0, 2], 2, L.FF_PARALLEL), ([1, 0, 1, 0], 2, L.FF_SEQ2SEQ), ([2, 2, 2], 3, L.FF_PARALLEL)):
        _, stop = ffd.apply_global_stop(torch.ones(1, len(counts) + 1, dtype=torch.int64), counts, n, v)
        assert stop == (ffd.stop_step(counts, n, v) or len(counts))
