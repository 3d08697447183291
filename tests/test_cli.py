"""CPU: the decode CLI's batching and multi-process paths (main.run_test) write the same files as the one-sample-at-a-time
single-process run.  The one step that needs the GPU -- main.decode_batch -- is replaced by the CPU oracle (tests only); the
GPU runs of the same driver live in test_parity_golden.py::test_cli_*."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(root):
    """Five small wireframes with different edge counts + the config / weights of a small parallel model."""
    sys.path.insert(0, ROOT)
    from faceformer_amd.config import load_cfg
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    os.makedirs(os.path.join(root, "json"), exist_ok=True)
    rng = np.random.default_rng(11)
    names = []
    for i in range(5):
        n = 7 + 2 * i
        raw = {"edges": [rng.uniform(-1, 1, size=(2, 2)).tolist() for _ in range(n)],
               "faces_indices": [[0, [[0, 1, 2]]], [1, [[3, 4, 5, 6]]]], "pairings": {}, "dominant_directions": [[1, 0, 0]]}
        with open(os.path.join(root, "json", "%08d.json" % i), "w") as f:
            json.dump(raw, f)
        names.append("json/%08d.json" % i)
    with open(os.path.join(root, "test.txt"), "w") as f:
        f.write("\n".join(names) + "\n")
    cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"),
                   ["model.num_lines", "16", "model.max_face_length", "8", "model.num_model", "128", "model.num_head", "2",
                    "model.num_feedforward", "256", "model.num_encoder_layers", "2", "model.num_decoder_layers", "2",
                    "root_dir", str(root), "post_process.is_coedge", "False"])
    sd = make_state_dict(state_dict_spec("parallel", 16, 8, 128, 256, 2, 2), "gain4", 3)
    return cfg, sd


def _oracle_decode(sd):
    from oracle import refpath

    def decode(_model, batch):
        return refpath.parallel_forward_eval(sd, batch, num_head=2)["predict"].numpy()
    return decode


def _files(d):
    return {n: open(os.path.join(d, n), "rb").read() for n in sorted(os.listdir(d))}


def _rank(rank, world, port, root, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import main as cli
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    cfg, sd = _cfg_only(root)
    cli.decode_batch = _oracle_decode(sd)
    cli.run_test(cfg, None, out_dir=out, device="cpu", batch_size=2, dist_mod=dist, model=object())
    dist.destroy_process_group()


def _cfg_only(root):
    sys.path.insert(0, ROOT)
    from faceformer_amd.config import load_cfg
    from faceformer_amd.synth import make_state_dict, state_dict_spec
    cfg = load_cfg(os.path.join(ROOT, "configs", "ours.yml"),
                   ["model.num_lines", "16", "model.max_face_length", "8", "model.num_model", "128", "model.num_head", "2",
                    "model.num_feedforward", "256", "model.num_encoder_layers", "2", "model.num_decoder_layers", "2",
                    "root_dir", str(root), "post_process.is_coedge", "False"])
    return cfg, make_state_dict(state_dict_spec("parallel", 16, 8, 128, 256, 2, 2), "gain4", 3)


def test_cli_batched_and_two_rank_runs_write_the_single_process_files(tmp_path):
    sys.path.insert(0, ROOT)
    import main as cli
    root = str(tmp_path / "data")
    cfg, sd = _setup(root)
    old = cli.decode_batch
    cli.decode_batch = _oracle_decode(sd)
    try:
        one = cli.run_test(cfg, None, out_dir=str(tmp_path / "b1"), device="cpu", batch_size=1, model=object())
        three = cli.run_test(cfg, None, out_dir=str(tmp_path / "b3"), device="cpu", batch_size=3, model=object())
    finally:
        cli.decode_batch = old
    ref = _files(one)
    assert len(ref) == 5 and all(json.loads(v)["pred_faces"] is not None for v in ref.values())
    assert _files(three) == ref                       # micro-batched: byte-identical records
    # two ranks over gloo: contiguous shares (3 + 2 samples), records all-gathered, rank 0 writes
    ctx = mp.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "w2")
    procs = [ctx.Process(target=_rank, args=(r, 2, port, root, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert _files(out) == ref


def test_cli_parser_takes_batch_size_next_to_the_reference_flags():
    sys.path.insert(0, ROOT)
    from faceformer_amd.config import get_parser
    parser = get_parser()
    parser.add_argument("--batch-size", type=int, default=1)
    a = parser.parse_args(["--config-file", "configs/ours.yml", "--test_ckpt", "x.ckpt", "--batch-size", "16", "model.num_lines", "256"])
    assert a.batch_size == 16 and a.opts == ["model.num_lines", "256"] and a.test_ckpt == "x.ckpt"
