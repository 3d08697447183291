#!/usr/bin/env python
"""Thin decode driver with the reference CLI's flags (reference main.py:24-51, test branch only):

    python main.py --config-file configs/ours.yml --test_ckpt last.ckpt [KEY VALUE ...]

loads the checkpoint's weights into the MI355X-native model, decodes every sample of
`cfg.datasets_test` (batch size 1 like the reference's test loader, trainer.py:51) and writes one
JSON per sample (`edges`, `dominant_directions`, `pred_faces`, `label_faces`; trainer.py:118-136)
under logs/<name>/<version>/json/, printing the running mean decode time and precision / recall.
Training / validation / resume (Lightning) are out of scope of this build.
"""
import json
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


def run_test(cfg, ckpt_path, out_dir=None, device="cuda", limit=None):
    model_class = getattr(models, cfg.model_class)
    dataset_class = getattr(D, cfg.dataset_class)
    model = model_class(**cfg.model)
    sd, _ = load_lightning_checkpoint(ckpt_path)
    model.load_state_dict(sd)
    model = model.eval().to(device)
    ds = dataset_class(cfg.root_dir, cfg.datasets_test, cfg.model)
    out_dir = out_dir or os.path.join("logs", cfg.trainer.name, str(cfg.trainer.version), "json")
    os.makedirs(out_dir, exist_ok=True)
    parallel = cfg.model_class == "SurfaceFormer_Parallel"
    total, stats = 0.0, []
    for i in range(len(ds) if limit is None else min(limit, len(ds))):
        item = ds[i]
        batch = D.collate([item])
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            out = model(batch)
        torch.cuda.synchronize()
        total += time.time() - t0
        raw = ds.raw_datas[i]
        pred, lab = out["predict"][0].cpu().numpy(), item["label"]
        parse = FZ.parse_parallel_faces if parallel else FZ.parse_faces
        pf, lf = parse(pred, lab, len(raw["edges"]), cfg.model.token)
        if cfg.post_process.is_coedge:
            pairings = raw.get("pairings", {})
            tol = cfg.post_process.enclosedness_tol
            pf = FZ.postprocess_faces(pf, raw["edges"], pairings, tol)
            lf = FZ.postprocess_faces(lf, raw["edges"], pairings, tol)
        m = FZ.face_metrics(pf, lf)
        stats.append((m["precision"], m["recall"], m["type_acc"]))
        rec = FZ.faces_record(raw["edges"], raw.get("dominant_directions", []), m["predictions"], m["labels"])
        name = os.path.splitext(os.path.basename(item["name"]))[0]
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            f.write(FZ.dumps_record(rec))
        print("Avg Time", total / (i + 1), "seconds.")
    if stats:
        n = len(stats)
        print("test_precision %.4f test_recall %.4f test_type_acc %.4f over %d samples"
              % (sum(s[0] for s in stats) / n, sum(s[1] for s in stats) / n, sum(s[2] for s in stats) / n, n))
    return out_dir


if __name__ == "__main__":
    args = get_parser().parse_args()
    cfg = get_cfg(args)
    if args.test_ckpt == "":
        raise SystemExit("only --test_ckpt (greedy decode + JSON dump) is implemented; training, "
                         "validation and resume are out of scope of the MI355X decode build")
    run_test(cfg, args.test_ckpt)
