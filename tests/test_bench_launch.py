"""CPU: bench.py's launch contract -- `--gpus N` is what runs, or nothing does (missing #1)."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_dry_spawn_builds_a_two_rank_group_and_plans_the_shards():
    """`bench.py --gpus 2 --dry-spawn` re-executes itself through torch.distributed.run with two ranks on 127.0.0.1, builds a gloo
    group and agrees the shard sizes with the collectives of decode_sharded(local_shard=True): config 3's 128 wireframes per rank."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-spawn"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # stdout is the one line and nothing else: gloo's "[Gloo] Rank r is connected to ..." (written to fd 1 of every rank by its
    # native side when a group is made) is sent to stderr
    out = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and out[0].startswith("{"), p.stdout[:500]
    d = json.loads(out[0])
    assert d["dry_spawn"] is True and d["n_gpus"] == 2 and d["ranks_in_group"] == 2 and d["backend"] == "gloo"
    assert d["shard_sizes"] == [128, 128] and d["global_batch"] == 256 and d["global_F"] == 256


def test_gpus_n_without_n_devices_fails_loudly_instead_of_running_one_gpu():
    """No launcher, `--gpus 2`, fewer than two devices visible (this container has none; the one-GPU box has one): non-zero exit
    and NO JSON line -- never `n_gpus: 1` for a `--gpus 2` request."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices are visible here: the request would be honoured")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2" in p.stderr


def test_world_size_that_disagrees_with_gpus_is_refused():
    env = dict(_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-spawn"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_compact_line_of_the_round_4_record_is_parsable_and_small():
    """the driver could not parse round 4's 25 KB line.  bench.compact_line() of that very record must stay
    under 4 KB and still carry the contract keys, `roofline.frac` and `cpu_baseline.value`."""
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r04", "bench_r04_B.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 20000
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) / full["value"] < 1e-5
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5 and d["roofline"]["bound"] == "mfma"
    assert abs(d["roofline"]["traffic"] / full["roofline"]["traffic"] - 1) < 1e-5
    assert abs(d["cpu_baseline"]["value"] - full["cpu_baseline"]["value"]) < 1e-2 and len(d["cpu_baseline"]["sample"]) <= 160
    assert set(d["other_configs"]) == set(full["other_configs"])
    assert all(set(v) <= {"value", "ms_per_step", "frac", "kernel_frac", "bf16x3", "default"} for v in d["other_configs"].values())
    assert "workload" in d["config"] and "model" not in d["config"]


def test_compact_line_sheds_optional_blocks_rather_than_grow():
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r04", "bench_r04_B.json")) as f:
        full = json.load(f)
    full["other_configs"] = {"cfg%03d" % i: dict(full["other_configs"]["C128"]) for i in range(60)}
    line = bench.compact_line(full)
    d = json.loads(line)
    assert len(line) < 4096 and "other_configs" not in d and "roofline" in d and "cpu_baseline" in d
