"""Multi-GPU decode: wireframes are independent, so the batch is sharded by wireframe (one process per
GPU) and only RESULTS travel: one `all_gather_into_tensor` of the fixed-shape int32 token tensor over
RCCL/xGMI.  No collective sits on the data path of the kernels.

The reference has no distributed code; what it fixes is the meaning of the raw
`predict` tensor, which couples the wireframes of a batch in two places: the padded sequence count
F = max(num_input) over the WHOLE batch (reference model_para.py:187) and the stop rule, evaluated
over ALL sequences of the batch (model_para.py:232).  `decode_sharded` therefore (a) decodes every
shard with the global F, (b) runs all T-1 steps locally without applying a stop rule, (c) sums the
per-step special-token counters over ranks (tiny all_reduce) and applies the GLOBAL rule, and
(d) gathers.  The result is identical to the single-GPU tensor for any world size.

Which rank decodes which wireframe (`shard_plan`): equal-sized wireframes are dealt out in contiguous
blocks; a ragged batch (BASELINE config E: 64..1024 edges) is balanced by decode COST, which grows like
n^2 (n anchor sequences, each attending to n keys and paying n-independent projections per token):
longest-processing-time-first over the wireframes sorted by edge count.  Every rank computes the same
plan from `num_input`, so no extra communication is needed to reassemble the batch order.

A rank may also hold ONLY its own wireframes (`local_shard=True`): then F, the stop-rule counters and the
wireframe counts are agreed by collectives and the result is the concatenation in rank order.
"""
import os

import torch

from .hip import lib as _L


def shard_range(n_items, rank, world):
    """Contiguous block partition: [lo, hi) of `n_items` owned by `rank` (blocks of equal size
    ceil(n/world); trailing ranks may own fewer or none)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per), per


def wireframe_cost(n, F=None):
    """Relative decode cost of a wireframe with n real edges: its decoded sequences (n, plus one shared
    padding-anchor sequence when n < F) times a per-sequence cost with a fixed part (projections) and a
    part linear in the key count n + 4 (cross attention, pointer).  The constants come from the FLOP
    split of SURVEY.md Appendix E (per token-layer: 5.24 MFLOP of projections vs 4*S*E = 2048*S)."""
    seqs = n + (1 if (F is not None and n < F) else 0)
    return seqs * (5.24e6 + 2048.0 * (n + 4))


def shard_plan(num_input, world, F=None):
    """Assignment of wireframe indices to ranks: list (one entry per rank) of index lists, each sorted by
    descending edge count (the order the engine likes: tight micro-batches).  Uniform batches get the
    contiguous blocks of `shard_range`; ragged ones are balanced by `wireframe_cost` (LPT greedy)."""
    n_items = len(num_input)
    if len(set(int(n) for n in num_input)) <= 1:
        return [list(range(*shard_range(n_items, r, world)[:2])) for r in range(world)]
    order = sorted(range(n_items), key=lambda i: (-int(num_input[i]), i))
    load = [0.0] * world
    plan = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        plan[r].append(i)
        load[r] += wireframe_cost(int(num_input[i]), F)
    return plan


def gather_predictions(pred, dist_mod, group=None):
    """all_gather of an integer token tensor [n_local, ...] (same shape on every rank) ->
    [world * n_local, ...] int64 on every rank.  int32 on the wire (token ids < 2^31)."""
    world = dist_mod.get_world_size(group)
    send = pred.to(torch.int32).contiguous()
    out = torch.empty((world * send.size(0),) + tuple(send.shape[1:]), dtype=torch.int32, device=send.device)
    dist_mod.all_gather_into_tensor(out, send, group=group)
    return out.to(torch.int64)


def apply_global_stop(predict, counts, num_wireframes, variant):
    """Zero the tokens after the reference's stop step given GLOBAL per-step counters.
    predict [..., T]; counts[s] = #{tokens >= num_token} (parallel) or #{EOS} (seq2seq) at step s."""
    steps = len(counts)
    stop = steps
    if variant == _L.FF_PARALLEL:
        for s, c in enumerate(counts):
            if c == 0:
                stop = s + 1
                break
    else:
        cum = 0
        for s, c in enumerate(counts):
            cum += c
            if cum == num_wireframes:
                stop = s + 1
                break
    if stop + 1 < predict.size(-1):
        predict[..., stop + 1:] = 0
    return predict, stop


def stop_step(counts, num_wireframes, variant):
    """First step (1-based) at which the reference's loop breaks given GLOBAL per-step counters, or None."""
    if variant == _L.FF_PARALLEL:
        for s, c in enumerate(counts):
            if c == 0:
                return s + 1
        return None
    cum = 0
    for s, c in enumerate(counts):
        cum += c
        if cum == num_wireframes:
            return s + 1
    return None


_CONTROL_GROUPS = {}


def _control_group(dist_mod, group):
    """A process group whose collectives take HOST tensors (the periodic stop check sums a few integers per rank; it must not
    queue behind the decode steps on the device stream the way an RCCL collective would).  The group itself when its backend
    is gloo, otherwise a gloo twin of it, created once.

    Who has to call: `new_group` over the DEFAULT group is collective over every process, which holds whenever decode_sharded
    runs on the default group (every rank decodes).  For a strict SUB-group only its members reach this point, so the twin is
    made with `use_local_synchronization=True` (members only; torch >= 2.1) -- where that is not available the caller gets a
    RuntimeError here and decode_sharded falls back to applying the stop rule after the decode.  A caller may also hand its
    own host-side group to decode_sharded(control_group=...)."""
    if dist_mod.get_backend(group) == "gloo":
        return group
    # keyed by the group OBJECT (the default group's when `group` is None: it changes when the process group is re-initialised)
    key = (dist_mod, group if group is not None else getattr(getattr(dist_mod, "group", None), "WORLD", None))
    if key in _CONTROL_GROUPS and _CONTROL_GROUPS[key] is None:
        raise RuntimeError("not available (failed before)")
    if key not in _CONTROL_GROUPS:
        import datetime
        world_group = getattr(getattr(dist_mod, "group", None), "WORLD", None)
        whole_world = group is None or group is world_group or \
            dist_mod.get_world_size(group) == dist_mod.get_world_size()
        kw = {"backend": "gloo", "timeout": datetime.timedelta(seconds=CONTROL_TIMEOUT_S)}
        if whole_world:
            _CONTROL_GROUPS[key] = dist_mod.new_group(ranks=None, **kw)
        else:
            ranks = dist_mod.get_process_group_ranks(group)
            try:
                _CONTROL_GROUPS[key] = dist_mod.new_group(ranks=ranks, use_local_synchronization=True, **kw)
            except TypeError as e:    # a torch without member-only group creation: do not risk a hang in new_group
                raise RuntimeError("sub-group control twin needs torch.distributed.new_group(use_local_synchronization=True): %s" % e)
    return _CONTROL_GROUPS[key]


def ensure_control_group(dist_mod, group=None):
    """Create (or fetch) the host-side control group of decode_sharded NOW -- collective over the ranks of `group` like the first
    decode_sharded call would be.  For callers that own their stdout: gloo's C++ side prints "[Gloo] Rank r is connected to ..."
    on STDOUT of every rank when a group is made (bench.py prints exactly one line there and makes the group with fd 1 pointed at
    stderr)."""
    return _control_group(dist_mod, group)


# A rank that fails inside a stop check must not leave its peers blocked for ever -- but the first check of an early (or
# idle) rank also waits for its slowest peer to GET there (data loading, first-call library build, a very large shard), so the
# limit is torch's own 30 minutes unless the caller shortens it: FF_CONTROL_TIMEOUT_S, or dist.CONTROL_TIMEOUT_S before the
# first sharded decode.  A caller-supplied control_group keeps its own timeout.
CONTROL_TIMEOUT_S = int(os.environ.get("FF_CONTROL_TIMEOUT_S", "1800"))


def check_points(T, sync_every):
    """(enqueued steps, counted steps) at which ff_decode evaluates its stop rule: every sync_every steps, one period
    behind the steps it has enqueued (ff_engine.hip).  Ranks without wireframes replay this cadence."""
    return [(enq, enq - sync_every) for enq in range(2 * sync_every, T - 1, sync_every)] if sync_every > 0 else []


def _decode_local(model, sub, variant, T, F, num_input, extra_rows, stop_callback=None, sync_every=0, stop_each_eos=None):
    """Decode of the wireframes in `sub` with the batch-global F and WITHOUT the local stop rule (the batch-global one arrives
    through `stop_callback`, or afterwards) -> (predict [n, F, T], counts of the executed steps)."""
    parallel = variant == _L.FF_PARALLEL
    n = sub["input"].size(0)
    order = list(range(n))
    if parallel and getattr(model, "sort_by_edges", True) and extra_rows is None and len(set(num_input)) > 1:
        order = sorted(range(n), key=lambda i: -num_input[i])
        idx = torch.tensor(order, device=sub["input"].device)
        sub = {"input": sub["input"].index_select(0, idx), "input_mask": sub["input_mask"].index_select(0, idx)}
    eng, memory, mask, kv_len = model._encode(sub)
    out = eng.decode(memory, mask, kv_len, variant, T=T, F=F,
                     num_input=[num_input[i] for i in order] if parallel else None,
                     chunk_wireframes=model.chunk_wireframes, chunk_seqs=model.chunk_seqs,
                     chunk_max_seqs=getattr(model, "chunk_max_seqs", 0),
                     num_streams=model.num_streams, sync_every=sync_every if stop_callback else 0,
                     flags=model.decode_flags | (_L.FF_STOP_EACH_EOS if (not parallel and (
                         getattr(model, "stop_each_eos", False) if stop_each_eos is None else stop_each_eos)) else 0),
                     x3_min_rows=model.x3_min_rows, ln_fuse_max_rows=getattr(model, "ln_fuse_max_rows", 0), extra_mask=extra_rows,
                     tok_sos=model.token.SOS if not parallel else 1,
                     tok_eos=model.token.EOS if not parallel else 3, no_stop=stop_callback is None,
                     stop_callback=stop_callback)
    pred = out["predict"].view(n, F, T)
    if order != list(range(n)):
        inv = torch.empty(n, dtype=torch.long, device=pred.device)
        inv[torch.tensor(order, device=pred.device)] = torch.arange(n, device=pred.device)
        pred = pred.index_select(0, inv)
    return pred, out["step_counts"]


def decode_sharded(model, inputs, dist_mod, group=None, local_shard=False, control_group=None, stop_each_eos=None):
    """Decode a batch across the ranks of `group`; returns `inputs` with 'predict' [N, F, T] (parallel) /
    [N, T] (seq2seq) for the WHOLE batch on every rank, identical to a single-process
    `model(inputs)['predict']` of that batch.

    local_shard=False: `inputs` is the full batch (same dict on every rank); every rank decodes the
        wireframes `shard_plan` gives it and only those slices are touched on the device.
    local_shard=True : `inputs` holds only this rank's wireframes (any number, also zero rows); the batch is
        their concatenation in rank order.  Also returns inputs['shard_sizes'] (wireframes per rank).
    control_group: a host-side (gloo) process group over the same ranks for the in-decode stop checks; None = the twin that
        `_control_group` makes (collective over the ranks of `group`: EVERY rank of `group` must call decode_sharded).
    stop_each_eos: single-sequence model only -- None = the module's own `stop_each_eos`; True / False = this call's rule,
        passed down as an argument (the module attribute is NOT touched: other host threads may be decoding with the model)."""
    from .models import SurfaceFormer_Parallel
    rank, world = dist_mod.get_rank(group), dist_mod.get_world_size(group)
    parallel = isinstance(model, SurfaceFormer_Parallel)
    variant = _L.FF_PARALLEL if parallel else _L.FF_SEQ2SEQ
    T = model.max_face_length if parallel else model.num_labels
    dev = next(model.parameters()).device
    n_here = inputs["input"].size(0)
    num_input = [int(n) for n in inputs["num_input"]] if parallel else [0] * n_here
    extra = model._extra_mask(inputs) if hasattr(model, "_extra_mask") else None   # [n*F or n, S] uint8

    if local_shard:
        meta = torch.tensor([max(num_input) if (parallel and num_input) else 1, n_here, inputs["input"].size(1)],
                            dtype=torch.int64, device=dev)
        allmeta = torch.empty(3 * world, dtype=torch.int64, device=dev)
        dist_mod.all_gather_into_tensor(allmeta, meta, group=group)
        allmeta = allmeta.view(world, 3).tolist()
        F = max(m[0] for m in allmeta) if parallel else 1
        sizes = [int(m[1]) for m in allmeta]
        # every rank pads its wireframes to the SAME num_lines: F is the batch-global anchor count and a rank whose own
        # padded width were smaller than another rank's widest wireframe could not hold F anchors (ff_decode: F <= S)
        widths = sorted({int(m[2]) for m in allmeta if m[1] > 0})
        if len(widths) > 1:
            raise ValueError("decode_sharded(local_shard=True): ranks padded their wireframes to different num_lines %s; "
                             "pad every shard to the same width (cfg.model.num_lines)" % widths)
        mine = list(range(n_here))
        N = sum(sizes)
        if extra is not None and parallel and extra.size(0) != n_here * F:
            raise ValueError("extra_mask must have one row per (wireframe, anchor) of the GLOBAL width F=%d" % F)
    else:
        N = n_here
        F = max(num_input) if parallel else 1
        plan = shard_plan(num_input, world, F) if parallel else \
            [list(range(*shard_range(N, r, world)[:2])) for r in range(world)]
        mine = plan[rank]
        sizes = [len(p) for p in plan]
    per = max(max(sizes), 1)

    # The reference breaks its loop at the first step at which NO sequence of the batch selects an edge (parallel) / every
    # wireframe has produced its EOS (seq2seq).  No rank can see that alone, so every sync_every steps the ranks sum the
    # counters they have so far (host tensors on the control group: a few integers) and stop together; a rank without
    # wireframes takes part with zeros at the same check points.
    sync_every = int(getattr(model, "sharded_sync_every", getattr(model, "sync_every", 0)))   # 0: rule applied after the decode
    checks = sync_every > 0 and world > 1 and bool(check_points(T, sync_every))
    ctrl = None   # (None is also a valid handle: the default group)
    if checks and control_group is not None:
        ctrl = control_group
    elif checks:
        try:
            ctrl = _control_group(dist_mod, group)
        except (RuntimeError, ValueError) as e:   # no host-side backend in this build of torch.distributed: every rank
            import warnings                        # fails the same way and decodes all T - 1 steps, as before round 3
            warnings.warn("decode_sharded: no gloo control group (%s); the global stop rule is applied after the decode" % e)
            _CONTROL_GROUPS[(dist_mod, group if group is not None else getattr(getattr(dist_mod, "group", None), "WORLD", None))] = None
            checks = False

    def global_stop(my_counts):
        # the counted prefix only: every rank is at the same check point, so the lengths agree.  (A rank that fails here
        # leaves the decode; its peers' all-reduce then ends with the control group's timeout, CONTROL_TIMEOUT_S.)
        tot = torch.tensor(list(my_counts) or [0], dtype=torch.int64)
        dist_mod.all_reduce(tot, group=ctrl)
        return stop_step(tot[: len(my_counts)].tolist(), N, variant) is not None

    local = torch.zeros((per, F, T), dtype=torch.int64, device=dev)
    counts = torch.zeros(max(T - 1, 1), dtype=torch.int64, device=dev)
    if not mine and checks:
        for _enq, counted in check_points(T, sync_every):
            if global_stop([0] * counted):
                break
    if mine:
        if local_shard:
            sub, ni, ex = inputs, num_input, extra
        else:
            idx = torch.tensor(mine, device=inputs["input"].device)
            sub = {"input": inputs["input"].index_select(0, idx), "input_mask": inputs["input_mask"].index_select(0, idx)}
            ni = [num_input[i] for i in mine]
            ex = None
            if extra is not None:
                rows = extra.view(N, F if parallel else 1, -1).index_select(0, idx.to(extra.device))
                ex = rows.reshape(-1, extra.size(-1)).contiguous()
        pred, c = _decode_local(model, sub, variant, T, F, ni, ex, stop_callback=global_stop if checks else None,
                                sync_every=sync_every, stop_each_eos=stop_each_eos)
        local[: len(mine)] = pred
        counts[: len(c)] = torch.tensor(c, dtype=torch.int64, device=dev)
    dist_mod.all_reduce(counts, group=group)
    gathered = gather_predictions(local, dist_mod, group)          # [world * per, F, T]
    if local_shard:
        keep = [r * per + k for r in range(world) for k in range(sizes[r])]
        full = gathered[torch.tensor(keep, dtype=torch.long, device=gathered.device)] if keep else gathered[:0]
        inputs["shard_sizes"] = sizes
    else:
        full = torch.empty((N, F, T), dtype=torch.int64, device=gathered.device)
        src = [r * per + k for r in range(world) for k in range(sizes[r])]
        dst = [i for r in range(world) for i in plan[r]]
        if dst:
            full[torch.tensor(dst, dtype=torch.long, device=gathered.device)] = \
                gathered[torch.tensor(src, dtype=torch.long, device=gathered.device)]
    full, _ = apply_global_stop(full, counts[: T - 1].tolist(), N, variant)
    inputs["predict"] = full if parallel else full.view(N, T)
    return inputs


# ---- predicted face-loop JSON over the wire ---------------------------------------------------------
def gather_json_records(records, dist_mod, group=None, device=None):
    """All-gather variable-length JSON strings: every rank passes the list of records it owns and receives
    the concatenation over ranks in rank order.  Wire format: per rank a uint8 buffer of
    [u32 little-endian length | utf-8 bytes]* padded to the longest rank (two collectives: sizes, then one
    `all_gather_into_tensor` of the padded payload over RCCL/gloo)."""
    import struct
    world = dist_mod.get_world_size(group)
    blob = b"".join(struct.pack("<I", len(b)) + b for b in (r.encode("utf-8") for r in records))
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    dist_mod.all_gather_into_tensor(sizes, size, group=group)
    sizes = sizes.tolist()
    width = max(max(sizes), 1)
    send = torch.zeros(width, dtype=torch.uint8, device=dev)
    if blob:
        send[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    recv = torch.empty(world * width, dtype=torch.uint8, device=dev)
    dist_mod.all_gather_into_tensor(recv, send, group=group)
    raw = recv.cpu().numpy().tobytes()
    out = []
    for r in range(world):
        buf, pos = raw[r * width: r * width + sizes[r]], 0
        while pos < len(buf):
            (n,) = struct.unpack_from("<I", buf, pos)
            out.append(buf[pos + 4: pos + 4 + n].decode("utf-8"))
            pos += 4 + n
    return out


def decode_to_face_json(model, inputs, dist_mod, group=None, edges=None, dominant_directions=None,
                        pairings=None, is_coedge=False, tol=2e-4, local_shard=False):
    """decode_sharded + face parsing of the rank's own wireframes + JSON all-gather: every rank returns
    the list of N per-wireframe JSON records in batch order (`edges`, `dominant_directions`, `pred_faces`,
    `label_faces`; reference trainer.py:118-136).  `edges[i]` / `dominant_directions[i]` / `pairings[i]`
    are the raw-data entries of wireframe i when available (else the record carries empty lists).
    local_shard=True: `inputs` (and edges / dominant_directions / pairings) hold only this rank's wireframes; the records
    come back for the whole batch = the shards in rank order."""
    from . import faces as FZ
    from .models import SurfaceFormer_Parallel
    n_here = inputs["input"].size(0)
    parallel = isinstance(model, SurfaceFormer_Parallel)
    # seq2seq: the reference's batch rule counts a sample's repeated EOS too and can stop the batch before another sample's own
    # EOS; the records must be those of one-wireframe decodes, so the batch runs until EVERY wireframe has produced one
    # (the rule goes down as an ARGUMENT: toggling the module attribute would change what a concurrent model(inputs) on another
    #  host thread decodes with, and a rank that raised in between would leave its peers on a different rule)
    out = decode_sharded(model, inputs, dist_mod, group, local_shard=local_shard, stop_each_eos=None if parallel else True)
    rank, world = dist_mod.get_rank(group), dist_mod.get_world_size(group)
    if local_shard:
        sizes = out["shard_sizes"]
        N, base = sum(sizes), sum(sizes[:rank])
        plan = [list(range(sum(sizes[:r]), sum(sizes[:r + 1]))) for r in range(world)]
        mine = list(range(n_here))                     # indices into the rank's own inputs
        rows = [base + k for k in mine]                # ... and into the gathered predict
    else:
        N = n_here
        if parallel:
            ni = [int(n) for n in inputs["num_input"]]
            plan = shard_plan(ni, world, max(ni))
        else:
            plan = [list(range(*shard_range(N, r, world)[:2])) for r in range(world)]
        mine = plan[rank]
        rows = mine
    sel = torch.tensor(mine, dtype=torch.long)
    pred = out["predict"].cpu()[torch.tensor(rows, dtype=torch.long)].numpy() if mine else []
    labels = inputs["label"].cpu()[sel].numpy() if mine else []
    recs = []
    for k, i in enumerate(mine):
        n = int(inputs["num_input"][i]) if "num_input" in inputs else int((~inputs["input_mask"][i]).sum())
        fn = FZ.parse_parallel_faces if parallel else FZ.parse_faces
        # (parallel: the wireframe's own anchor sequences, not the batch-wide padding-anchor rows behind them)
        own = FZ.apply_own_stop_rule(pred[k][:n] if parallel else pred[k], model.token, parallel)   # as a one-sample decode leaves them
        pf, lf = fn(own, labels[k], n, model.token)
        if is_coedge and edges is not None:
            pf = FZ.postprocess_faces(pf, edges[i], pairings[i] if pairings else {}, tol)
            lf = FZ.postprocess_faces(lf, edges[i], pairings[i] if pairings else {}, tol)
        m = FZ.face_metrics(pf, lf)
        recs.append(FZ.dumps_record(FZ.faces_record(
            edges[i] if edges is not None else [], dominant_directions[i] if dominant_directions is not None else [],
            m["predictions"], m["labels"])))
    dev = out["predict"].device if dist_mod.get_backend(group) == "nccl" else torch.device("cpu")
    flat = gather_json_records(recs, dist_mod, group, device=dev)        # rank order
    order = [i for r in range(world) for i in plan[r]]
    res = [None] * N
    for rec, i in zip(flat, order):
        res[i] = rec
    return res
