"""Multi-GPU decode: wireframes are independent, so the batch is sharded by wireframe index (one
process per GPU) and only RESULTS travel: one `all_gather_into_tensor` of the fixed-shape int32
token tensor over RCCL/xGMI.  No collective sits on the data path of the kernels.

The reference has no distributed code; what it fixes is the meaning of the raw
`predict` tensor, which couples the wireframes of a batch in two places: the padded sequence count
F = max(num_input) over the WHOLE batch (reference model_para.py:187) and the stop rule, evaluated
over ALL sequences of the batch (model_para.py:232).  `decode_sharded` therefore (a) decodes every
shard with the global F, (b) runs all T-1 steps locally without applying a stop rule, (c) sums the
per-step special-token counters over ranks (tiny all_reduce) and applies the GLOBAL rule, and
(d) gathers.  The result is identical to the single-GPU tensor for any world size.
"""
import torch

from .hip import lib as _L


def shard_range(n_items, rank, world):
    """Contiguous block partition: [lo, hi) of `n_items` owned by `rank` (blocks of equal size
    ceil(n/world); trailing ranks may own fewer or none)."""
    per = (n_items + world - 1) // world
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per), per


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


def decode_sharded(model, inputs, dist_mod, group=None):
    """Decode the FULL batch `inputs` (same dict on every rank) across the ranks of `group`.
    Returns the dict with 'predict' [N, F, T] (parallel) / [N, T] (seq2seq) for the whole batch on
    every rank, identical to a single-process `model(inputs)['predict']`."""
    from .models import SurfaceFormer_Parallel
    rank, world = dist_mod.get_rank(group), dist_mod.get_world_size(group)
    N = inputs["input"].size(0)
    lo, hi, per = shard_range(N, rank, world)
    parallel = isinstance(model, SurfaceFormer_Parallel)
    variant = _L.FF_PARALLEL if parallel else _L.FF_SEQ2SEQ
    T = model.max_face_length if parallel else model.num_labels
    num_input = [int(n) for n in inputs["num_input"]] if parallel else None
    F = max(num_input) if parallel else 1
    dev = next(model.parameters()).device
    local = torch.zeros((per, F, T), dtype=torch.int64, device=dev)
    counts = torch.zeros(max(T - 1, 1), dtype=torch.int64, device=dev)
    if hi > lo:
        sub = {"input": inputs["input"][lo:hi], "input_mask": inputs["input_mask"][lo:hi]}
        eng, memory, mask, kv_len = model._encode(sub)
        out = eng.decode(memory, mask, kv_len, variant, T=T, F=F,
                         num_input=num_input[lo:hi] if parallel else None,
                         chunk_wireframes=model.chunk_wireframes, chunk_seqs=model.chunk_seqs,
                         num_streams=model.num_streams, sync_every=0, flags=model.decode_flags,
                         tok_sos=model.token.SOS if not parallel else 1,
                         tok_eos=model.token.EOS if not parallel else 3, no_stop=True)
        local[: hi - lo] = out["predict"].view(hi - lo, F, T)
        c = out["step_counts"]
        counts[: len(c)] = torch.tensor(c, dtype=torch.int64, device=dev)
    dist_mod.all_reduce(counts, group=group)
    full = gather_predictions(local, dist_mod, group)[:N]
    full, _ = apply_global_stop(full, counts[: T - 1].tolist(), N, variant)
    inputs["predict"] = full if parallel else full.view(N, T)
    return inputs


# ---- predicted face-loop JSON over the wire ---------------------------------------------------------
def gather_json_records(records, dist_mod, group=None, device=None):
    """All-gather variable-length JSON strings: every rank passes the list of records it owns (in
    wireframe order) and receives the concatenation over ranks in rank order.  Wire format: per rank a
    uint8 buffer of [u32 little-endian length | utf-8 bytes]* padded to the longest rank (two
    collectives: sizes, then one `all_gather_into_tensor` of the padded payload over RCCL/gloo)."""
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
                        pairings=None, is_coedge=False, tol=2e-4):
    """decode_sharded + face parsing of the rank's own wireframes + JSON all-gather: every rank returns
    the list of N per-wireframe JSON records (`edges`, `dominant_directions`, `pred_faces`,
    `label_faces`; reference trainer.py:118-136).  `edges[i]` / `dominant_directions[i]` / `pairings[i]`
    are the raw-data entries of wireframe i when available (else the record carries empty lists)."""
    from . import faces as FZ
    from .models import SurfaceFormer_Parallel
    out = decode_sharded(model, inputs, dist_mod, group)
    rank, world = dist_mod.get_rank(group), dist_mod.get_world_size(group)
    N = inputs["input"].size(0)
    lo, hi, _ = shard_range(N, rank, world)
    parallel = isinstance(model, SurfaceFormer_Parallel)
    pred = out["predict"][lo:hi].cpu().numpy()
    labels = inputs["label"][lo:hi].cpu().numpy()
    recs = []
    for k, i in enumerate(range(lo, hi)):
        n = int(inputs["num_input"][i]) if "num_input" in inputs else int((~inputs["input_mask"][i]).sum())
        fn = FZ.parse_parallel_faces if parallel else FZ.parse_faces
        pf, lf = fn(pred[k], labels[k], n, model.token)
        if is_coedge and edges is not None:
            pf = FZ.postprocess_faces(pf, edges[i], pairings[i] if pairings else {}, tol)
            lf = FZ.postprocess_faces(lf, edges[i], pairings[i] if pairings else {}, tol)
        m = FZ.face_metrics(pf, lf)
        recs.append(FZ.dumps_record(FZ.faces_record(
            edges[i] if edges is not None else [], dominant_directions[i] if dominant_directions is not None else [],
            m["predictions"], m["labels"])))
    dev = out["predict"].device if dist_mod.get_backend(group) == "nccl" else torch.device("cpu")
    return gather_json_records(recs, dist_mod, group, device=dev)
