#!/usr/bin/env python
"""Thin decode driver with the reference CLI's flags (reference main.py:24-51, test branch only):

    python main.py --config-file configs/ours.yml --test_ckpt last.ckpt [--batch-size N] [KEY VALUE ...]
    python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 main.py ...   (one rank per GPU)

loads the checkpoint's weights into the MI355X-native model, decodes every sample of `cfg.datasets_test` and
writes one JSON per sample (`edges`, `dominant_directions`, `pred_faces`, `label_faces`; trainer.py:118-136)
under logs/<name>/<version>/json/, printing the running mean decode time and precision / recall.

--batch-size N (extension; the reference's test loader is fixed at 1, trainer.py:51): N samples per `model(batch)`
call, i.e. the micro-batched engine -- 128 wireframes per call run at 0.84 of the f32 matrix peak, one at 0.63
(DESIGN.md 5).  The records do not depend on N: a wireframe's faces are parsed from its OWN anchor rows (not the
batch-wide padding-anchor rows behind them) up to its OWN stop step (faces.apply_own_stop_rule: in a batch the
loop runs on until every wireframe is done), i.e. from exactly the tokens a one-sample decode leaves.  For the
single-sequence model that needs another batch rule than the reference's: its loop stops when the CUMULATIVE number
of EOS tokens equals the batch size (model.py:207-210), which a sample that repeats its EOS reaches before another
sample has produced its own; with N > 1 the decode therefore runs until EVERY wireframe has produced an EOS
(SurfaceFormer.stop_each_eos, ff_decode flag FF_STOP_EACH_EOS).

Under torch.distributed.run every rank decodes a contiguous share of the samples on its own GPU; the JSON
records are all-gathered (faceformer_amd.dist.gather_json_records: RCCL on GPUs, gloo on CPU) and rank 0
writes the files, so a G-rank run leaves exactly the files of a single-process run.
Training / validation / resume (Lightning) are out of scope of this build.
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from faceformer_amd import datasets as D  # noqa: E402
from faceformer_amd import faces as FZ  # noqa: E402
from faceformer_amd import models  # noqa: E402
from faceformer_amd.checkpoint import load_lightning_checkpoint  # noqa: E402
from faceformer_amd.config import get_cfg, get_parser  # noqa: E402
from faceformer_amd.dist import gather_json_records, shard_range  # noqa: E402


def decode_batch(model, batch):
    """`model(batch)['predict']` as a numpy array (the one step of run_test that needs the GPU)."""
    with torch.no_grad():
        return model(batch)["predict"].cpu().numpy()


def record_of(cfg, raw, item, pred, parallel):
    """(JSON text, (precision, recall, type accuracy)) of one decoded sample (reference trainer.py:118-136, 210-300)."""
    parse = FZ.parse_parallel_faces if parallel else FZ.parse_faces
    if parallel:
        # the wireframe's OWN anchor sequences: in a batch `predict` is padded to F = max(num_input) rows per wireframe with
        # padding-anchor sequences (reference model_para.py:204-205), which the reference's one-sample test batches never contain
        pred = pred[: int(item["num_input"])]
    # ... and their tokens up to the wireframe's OWN stop step (in a batch the loop runs on until every wireframe is done)
    pred = FZ.apply_own_stop_rule(pred, cfg.model.token, parallel)
    pf, lf = parse(pred, item["label"], len(raw["edges"]), cfg.model.token)
    if cfg.post_process.is_coedge:
        pairings = raw.get("pairings", {})
        tol = cfg.post_process.enclosedness_tol
        pf = FZ.postprocess_faces(pf, raw["edges"], pairings, tol)
        lf = FZ.postprocess_faces(lf, raw["edges"], pairings, tol)
    m = FZ.face_metrics(pf, lf)
    rec = FZ.faces_record(raw["edges"], raw.get("dominant_directions", []), m["predictions"], m["labels"])
    return FZ.dumps_record(rec), (m["precision"], m["recall"], m["type_acc"])


def run_test(cfg, ckpt_path, out_dir=None, device="cuda", limit=None, batch_size=1, dist_mod=None, model=None):
    """Decode cfg.datasets_test and write the per-sample JSON files; returns the output directory.
    dist_mod: an initialised torch.distributed (or None): the samples are sharded over its ranks, the records gathered,
    rank 0 writes.  model: a ready model object (tests), else built from cfg + checkpoint."""
    model_class = getattr(models, cfg.model_class)
    dataset_class = getattr(D, cfg.dataset_class)
    if model is None:
        model = model_class(**cfg.model)
        sd, _ = load_lightning_checkpoint(ckpt_path)
        model.load_state_dict(sd)
        model = model.eval().to(device)
    ds = dataset_class(cfg.root_dir, cfg.datasets_test, cfg.model)
    out_dir = out_dir or os.path.join("logs", cfg.trainer.name, str(cfg.trainer.version), "json")
    parallel = cfg.model_class == "SurfaceFormer_Parallel"
    batch_size = max(1, int(batch_size))
    if not parallel and batch_size > 1 and hasattr(model, "stop_each_eos"):
        model.stop_each_eos = True      # every wireframe decodes up to its own EOS (module docstring)
    rank = dist_mod.get_rank() if dist_mod is not None else 0
    world = dist_mod.get_world_size() if dist_mod is not None else 1
    n_all = len(ds) if limit is None else min(limit, len(ds))
    lo, hi, _ = shard_range(n_all, rank, world)
    batch_size = max(1, int(batch_size))
    total, done, stats, records = 0.0, 0, [], []
    for b0 in range(lo, hi, batch_size):
        idx = list(range(b0, min(hi, b0 + batch_size)))
        items = [ds[i] for i in idx]
        batch = D.collate(items)
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        if torch.cuda.is_available() and str(device).startswith("cuda"):
            torch.cuda.synchronize()
        t0 = time.time()
        pred = decode_batch(model, batch)
        total += time.time() - t0
        done += len(idx)
        for k, i in enumerate(idx):
            text, st = record_of(cfg, ds.raw_datas[i], items[k], pred[k], parallel)
            stats.append(st)
            records.append((os.path.splitext(os.path.basename(items[k]["name"]))[0], text))
        print("Avg Time", total / done, "seconds.")
    if dist_mod is not None and world > 1:
        # names and records travel the same way (two gathers of length-prefixed utf-8); rank order = sample order
        dev = torch.device(device) if dist_mod.get_backend() == "nccl" else None
        names = gather_json_records([n for n, _ in records], dist_mod, device=dev)
        texts = gather_json_records([t for _, t in records], dist_mod, device=dev)
        flat = [v for s in stats for v in s]
        allstats = [None] * world
        dist_mod.all_gather_object(allstats, flat)
        stats = [tuple(fl[i:i + 3]) for fl in allstats for i in range(0, len(fl), 3)]
        records = list(zip(names, texts))
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        for name, text in records:
            with open(os.path.join(out_dir, name + ".json"), "w") as f:
                f.write(text)
        if stats:
            n = len(stats)
            print("test_precision %.4f test_recall %.4f test_type_acc %.4f over %d samples"
                  % (sum(s[0] for s in stats) / n, sum(s[1] for s in stats) / n, sum(s[2] for s in stats) / n, n))
    if dist_mod is not None and world > 1:
        dist_mod.barrier()
    return out_dir


if __name__ == "__main__":
    parser = get_parser()
    parser.add_argument("--batch-size", type=int, default=1,
                        help="samples per model(batch) call (the reference's test loader is fixed at 1); the records do not depend on it")
    args = parser.parse_args()
    cfg = get_cfg(args)
    if args.test_ckpt == "":
        raise SystemExit("only --test_ckpt (greedy decode + JSON dump) is implemented; training, "
                         "validation and resume are out of scope of the MI355X decode build")
    dist_mod, device = None, "cuda"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist_mod
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        device = "cuda:%d" % local_rank
        dist_mod.init_process_group("nccl", device_id=torch.device(device))
    run_test(cfg, args.test_ckpt, device=device, batch_size=args.batch_size, dist_mod=dist_mod)
    if dist_mod is not None:
        dist_mod.destroy_process_group()
