#!/usr/bin/env python
"""Throughput of the greedy pointer-decode path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

Metric (BASELINE.json): decoded edges/s = pointer selections per second on 256-edge wireframes.
One "step" = one pass of the hot path over one batch per GPU: encoder + all greedy decode steps +
output packing (+ the RCCL all-gather of the predicted loops when N > 1).

N = 1 (the headline): BASELINE config 2 ("B"): configs/ours.yml with model.num_lines=256, ONE synthetic 256-edge
wireframe (F=256 anchor sequences x 36 steps = 9216 selections), default-xavier synthetic weights (never stops early),
fp32, every product on the f32 matrix cores.  The same run also measures, with a few timed passes each, the other
BASELINE configurations that fit one GPU and reports them under `other_configs` (each with its own roofline):
    C128  config 3's per-GPU share: 128 of the 1024 256-edge wireframes                    (configs/ours.yml)
    E32   config 5's per-GPU share: 32 ragged 64..1024-edge wireframes                     (configs/ours-perspective.yml)
    D     config 4: seq2seq+coedge.yml, one 216-edge wireframe, extra pointer mask operand (SurfaceFormer, 258 steps)
    A     config 1's sizes: seq2seq.yml, one 64-edge wireframe                             (SurfaceFormer, 258 steps)

N > 1: BASELINE config 3 itself: every rank decodes its share of the 1024-wireframe batch -- 1024/8 = 128 wireframes per
GPU (at N < 8 the batch is 128 N wireframes: weak scaling of config 3's per-GPU workload) -- through
`faceformer_amd.dist.decode_sharded(local_shard=True)`: batch-global F, no local stop, all-reduced stop counters, the
GLOBAL stop rule, RCCL all-gather of the int32 tokens inside the timed step.  The N = 1 reference of that series is
`other_configs.C128` of the N = 1 line.  `weak_one_wireframe_per_gpu` carries the one-wireframe-per-GPU line as well.

The JSON line also carries
  roofline     : the dominant kernel family (the f32-MFMA GEMM): algorithmic flops (2MNK summed over its launches of one
                 step) / its summed duration, measured with HIP events on the launch stream by the library's profiling
                 hooks, NET of the event bracket (ff_profile_bracket_us: the interval an event pair reports around an
                 empty kernel), against the 157.3 TF/s f32 matrix peak;
  cpu_baseline : the CPU oracle (op-for-op restatement of the reference, oracle/refpath.py) timed on min(physical cores, 32)
                 host threads on a bounded sample of the workload: the first 128 anchor sequences of ONE wireframe, all steps
                 (~22 s; --cpu-anchors 256 = the full wireframe; if it does not finish within --cpu-timeout, 32 anchors;
                 sequences are independent, so the sample is faithful) -- `sample` says which; `first_divergence` names the first (sequence, step) at
                 which the GPU's tokens leave the oracle's, with the oracle's own top-2 margin there and the tolerance.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_HBM_TBS = 8.0            # MI355X_MICROARCH.md: HBM3E
NOMINAL_GHZ = 2.4             # the clock both matrix peaks are quoted at
E_SIZES, E_PROBS, E_SEED = (64, 128, 256, 512, 1024), (.3, .3, .2, .1, .1), 2024
CAT_NAMES = ["gemm_f32_kernels", "attention_kernels", "layernorm_kernel", "pointer_kernels", "row_ops", "unused",
             "gemm_bf16x3_kernel"]
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense; the split product spends 6 of them per fp32 product


def alg_flops_per_wireframe(n, T, E=512, FF=1024, layers=6, in_dim=100, F=None):
    """algorithmic flops of one wireframe (parallel model: F = n sequences; seq2seq: F = 1)."""
    S, steps = n + 4, T - 1
    F = n if F is None else F
    st = steps * (steps + 1) // 2
    st2 = steps * (steps + 1) * (2 * steps + 1) // 6
    embed = n * 2 * (in_dim * E + E * E)
    enc = layers * (S * 2 * (4 * E * E + 2 * E * FF) + 4 * S * S * E)
    cross_kv = layers * S * 2 * (2 * E * E)
    dec_lin = layers * F * st * 2 * (6 * E * E + 2 * E * FF)
    dec_self = layers * F * 4 * E * st2
    dec_cross = layers * F * 4 * S * E * st
    ptr = F * steps * (2 * E * E + 2 * S * E)
    return embed + enc + cross_kv + dec_lin + dec_self + dec_cross + ptr


def decoder_weight_bytes(E=512, FF=1024, layers=6):
    """fp32 bytes of the decoder stack + project (what a seq2seq decode step streams, SURVEY 8d: 76.8 MB)."""
    per_layer = 2 * (3 * E * E + 3 * E + E * E + E) + 2 * E * FF + FF + E + 6 * E
    return 4 * (layers * per_layer + 2 * E + E * E + E)


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def config_e_edge_counts(total):
    """Seed-listed edge counts of BASELINE config 5: n in {64..1024} with p = {.3,.3,.2,.1,.1}."""
    import numpy as np
    return [int(v) for v in np.random.default_rng(E_SEED).choice(E_SIZES, size=total, p=E_PROBS)]


def run_cpu_child(code, timeout, threads):
    import subprocess
    cp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout,
                        env=dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads)))
    return json.loads(cp.stdout.strip().splitlines()[-1])


def timed(step_fn, fence, warmup, steps):
    for _ in range(warmup):
        step_fn()
    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
        out = step_fn()
    fence()
    return time.perf_counter() - t0, out


def profile_once(lib, L, run_once):
    """One pass under the library's event hooks -> (ms, work, launches, algorithmic bytes) per category."""
    ncat = len(CAT_NAMES)
    ms, work, cnt = (ctypes.c_double * ncat)(), (ctypes.c_double * ncat)(), (ctypes.c_longlong * ncat)()
    torch.cuda.synchronize()
    lib.ff_profile_begin()
    run_once()
    L.check(lib.ff_profile_end(ms, work, cnt, ncat), "ff_profile_end")
    alg_bytes = (ctypes.c_double * ncat)()
    L.check(lib.ff_profile_bytes(alg_bytes, ncat), "ff_profile_bytes")
    return list(ms), list(work), [int(c) for c in cnt], list(alg_bytes)


def gemm_roofline(prof, wall_ms, empty_us, traffic=None, traffic_src=None):
    """`roofline` object of the dominant kernel family from one profiled pass.  The event bracket inflates every measured
    interval; it is calibrated PER CONFIGURATION as (sum of the bracketed intervals - un-instrumented wall time of one pass)
    / launches: the decode queue never drains (kernel trace under profiles/), so the kernels' own durations sum to the wall
    time.  Times reported are net of it; the raw event sums are kept beside them."""
    ms, work, cnt, alg_bytes = prof
    nl = max(1, sum(cnt))
    bracket_us = max(0.0, (sum(ms) - wall_ms) / nl * 1e3)
    net = [max(0.0, ms[i] - cnt[i] * bracket_us * 1e-3) for i in range(len(ms))]
    total_net = sum(net)
    ach = work[0] / (net[0] * 1e-3) / 1e12 if net[0] > 0 else 0.0
    ach_gross = work[0] / (ms[0] * 1e-3) / 1e12 if ms[0] > 0 else 0.0
    roof = {
        "kernel": "f32-MFMA GEMM (ff_gemm.hip; all launch shapes of one tiling family)",
        "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
        "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
        "alg_bytes_per_launch": alg_bytes[0] / max(1, cnt[0]), "launches_per_step": cnt[0],
        "avg_launch_us": 1e3 * net[0] / max(1, cnt[0]), "alg_flop_per_launch": work[0] / max(1, cnt[0]),
        "share_of_kernel_time": net[0] / total_net if total_net > 0 else None,
        "event_bracket_us_per_launch": bracket_us, "event_interval_of_an_empty_kernel_us": empty_us,
        "achieved_with_bracket": ach_gross, "frac_with_bracket": ach_gross / PEAK_F32_MFMA_TFLOPS,
        "note": "achieved / avg_launch_us are NET of the event bracket = (sum of bracketed intervals - un-instrumented pass time) / "
                "launches of this configuration (the queue never drains, so kernel durations sum to the pass time); "
                "*_with_bracket are the raw event sums; the rocprofv3 kernel trace under profiles/ is the bracket-free figure",
    }
    extra = {
        "kernel_time_ms_per_step": {CAT_NAMES[i]: net[i] for i in range(len(ms))},
        "kernel_time_ms_per_step_with_bracket": {CAT_NAMES[i]: ms[i] for i in range(len(ms))},
        "kernel_launches_per_step": {CAT_NAMES[i]: cnt[i] for i in range(len(ms))},
    }
    if net[1] > 0:
        extra["attention_tflops"] = work[1] / (net[1] * 1e-3) / 1e12
    return roof, extra


def live_traffic(timeout_s=150):
    """HBM bytes per launch of the f32 GEMM family, MEASURED IN THIS RUN: two child passes of the same workload under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains mixed in -- MI355X_MICROARCH.md's recipe),
    summed over the family's dispatches: (2 x FETCH_SIZE + WRITE_SIZE) KB / dispatches (FETCH_SIZE under-reports wide coalesced
    reads by 2x on gfx950).  Returns (bytes per launch, source text) or (None, reason)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    sums = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--steps", "1",
                   "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-x3-line", "--no-other-configs", "--no-live-traffic"]
            try:
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp",
                                    env=dict(os.environ, TMPDIR="/tmp"))
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s: no result within %d s" % (counter, timeout_s)
            dbs = [os.path.join(r, f) for r, _d, fs in os.walk(d) for f in fs if f.endswith(".db")]
            if cp.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, cp.returncode)
            try:
                rows = sqlite3.connect(dbs[0]).execute(
                    "select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
            except sqlite3.Error as e:
                return None, "rocpd database of the %s pass: %s" % (counter, e)
            vals = [float(v) for k, v in rows if "gemm_" in k and "x3" not in k]
            if not vals:
                return None, "no GEMM dispatches in the %s pass" % counter
            sums[counter] = (sum(vals), len(vals))
    f = sums["FETCH_SIZE"][0] / sums["FETCH_SIZE"][1]
    w = sums["WRITE_SIZE"][0] / sums["WRITE_SIZE"][1]
    return (2.0 * f + w) * 1024.0, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate child passes of the "
                                    "same workload, %d GEMM dispatches, (2 x FETCH_SIZE + WRITE_SIZE) KB per dispatch" % sums["FETCH_SIZE"][1])


def _committed_traffic(fname):
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", fname)))
    if not cands:
        return None, None
    with open(cands[-1]) as f:
        return json.load(f)["hbm_bytes_per_launch"], os.path.relpath(cands[-1], ROOT)


def x3_roofline(prof, wall_ms, with_traffic=False, kind="fp16x2"):
    """`roofline` of the split projection kernel (the package default's dominant kernel) from one profiled pass: algorithmic
    fp32 flops (2MNK) of its launches / their summed duration net of the event bracket, against the 16-bit matrix peak divided
    by the partial products an fp32 product costs: six with three bf16 terms, three with two fp16 terms (round 6)."""
    ms, work, cnt, alg_bytes = prof
    nl = max(1, sum(cnt))
    bracket_us = max(0.0, (sum(ms) - wall_ms) / nl * 1e3)
    net = [max(0.0, ms[i] - cnt[i] * bracket_us * 1e-3) for i in range(len(ms))]
    i3 = CAT_NAMES.index("gemm_bf16x3_kernel")
    nprod = 3.0 if kind == "fp16x2" else 6.0
    peak = PEAK_BF16_MFMA_TFLOPS / nprod
    ach = work[i3] / (net[i3] * 1e-3) / 1e12 if net[i3] > 0 else 0.0
    return {
        "kernel": "gemm_x3_kernel (ff_gemm_x3.hip): fp32-accurate product as %d %s MFMA partial products per K slice"
                  % (int(nprod), "fp16" if kind == "fp16x2" else "bf16"),
        "split_kind": kind,
        "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s (fp32-equivalent)", "frac": ach / peak,
        "traffic": _committed_traffic("traffic_x3.json")[0] if with_traffic else None,
        "traffic_unit": "HBM bytes per launch", "traffic_source": _committed_traffic("traffic_x3.json")[1] if with_traffic else None,
        "alg_bytes_per_launch": alg_bytes[i3] / max(1, cnt[i3]), "launches_per_step": cnt[i3],
        "avg_launch_us": 1e3 * net[i3] / max(1, cnt[i3]), "alg_flop_per_launch": work[i3] / max(1, cnt[i3]),
        "share_of_kernel_time": net[i3] / sum(net) if sum(net) > 0 else None,
        "event_bracket_us_per_launch": bracket_us,
        "kernel_time_ms_per_step": {CAT_NAMES[i]: net[i] for i in range(len(ms))},
        "kernel_launches_per_step": {CAT_NAMES[i]: cnt[i] for i in range(len(ms))},
        "note": "peak = 2500 TF/s dense 16-bit MFMA / %d partial products; the six-product bf16 form is POWER-limited on this chip "
                "(245-272 TF/s-equivalent on zero-filled operands, 170-200 on random ones: profiles/r04/x3v2_*.txt), the three-product "
                "fp16 form runs 226-245 on random operands (profiles/r06/gemm_split_kinds.txt); effective_clock_ghz = shader clock "
                "measured under this configuration's passes" % int(nprod),
    }


def path_roofline(falg, sec_per_step):
    return {"alg_tflop_per_gpu_step": falg / 1e12, "achieved_tflops_per_gpu": falg / sec_per_step / 1e12,
            "frac_of_f32_mfma_peak": falg / sec_per_step / 1e12 / PEAK_F32_MFMA_TFLOPS}


def _sig(v, n=6):
    """Floats to n significant digits (the compact line only; bench_detail.json keeps full precision)."""
    if isinstance(v, float):
        return float("%.*g" % (n, v))
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def _cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 3] + "..."


ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "launches_per_step",
             "avg_launch_us", "share_of_kernel_time", "effective_clock_ghz", "frac_at_effective_clock")
LINE_LIMIT = 4096


def compact_line(result):
    """The ONE line the driver parses (last line of stdout), kept under LINE_LIMIT bytes: the contract keys, a compact
    `roofline` and `cpu_baseline`, and {value, ms_per_step, frac} for the package default and each other configuration.
    Notes, bracket variants, per-category tables and the full `other_configs` live in bench_detail.json (and on
    the stderr line tagged "bench_detail")."""
    out = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                         "vs_baseline", "dtype", "data"))
    cfg = dict(result.get("config") or {})
    cfg["workload"] = _cut(cfg.get("workload"), 640)
    out["config"] = cfg
    if "path_roofline" in result:
        out["path_roofline"] = result["path_roofline"]
    if result.get("roofline"):
        r = _pick(result["roofline"], ROOF_KEYS)
        r["kernel"] = _cut(r.get("kernel"), 80)
        src = result["roofline"].get("traffic_source") or ""
        r["traffic_measured"] = "this run (rocprofv3 --pmc, 2 child passes)" if src.startswith("measured in this run") else _cut(src, 90)
        out["roofline"] = r
    if "kernel_time_ms_per_step" in result:
        out["kernel_time_ms_per_step"] = result["kernel_time_ms_per_step"]
    cb = result.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample", "tokens_identical_to_gpu", "divergent_sequences",
                       "threads_sweep"))
        c["sample"] = _cut(c.get("sample"), 160)
        out["cpu_baseline"] = c
    if "speedup_vs_cpu" in result:
        out["speedup_vs_cpu"] = result["speedup_vs_cpu"]

    def brief(e):
        b = _pick(e, ("value", "ms_per_step"))
        pr = e.get("path_roofline") or {}
        b["frac"] = pr.get("frac_of_f32_mfma_peak")
        rf = e.get("roofline") or {}
        if rf.get("frac") is not None:
            b["kernel_frac"] = rf["frac"]
        return b
    x3 = result.get("package_default")
    if x3:
        out["package_default"] = brief(x3)
        out["package_default"]["split_kind"] = x3.get("split_kind")
        if (x3.get("roofline") or {}).get("effective_clock_ghz") is not None:
            out["package_default"]["effective_clock_ghz"] = x3["roofline"]["effective_clock_ghz"]
    if result.get("other_configs"):
        oc = {}
        for name, e in result["other_configs"].items():
            oc[name] = brief(e)
            if e.get("package_default"):
                oc[name]["default"] = _pick(e["package_default"], ("value", "ms_per_step"))
        out["other_configs"] = oc
    for k in ("rehearsal", "rccl_ranks", "collective_backend", "rccl_version", "wireframes_per_s", "bench_seconds"):
        if k in result:
            out[k] = result[k]
    for k in ("weak_one_wireframe_per_gpu", "face_json_gather"):
        if k in result:
            out[k] = _pick(result[k], ("value", "ms_per_step", "steps"))
    if "scaling_series" in result:
        out["scaling_series"] = {k: _cut(v, 120) for k, v in result["scaling_series"].items() if k != "note"}
    out["detail"] = "bench_detail.json beside bench.py (also gpurun_out/bench_detail.json and the stderr line tagged bench_detail)"
    out = _sig(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= LINE_LIMIT:    # never let the driver's record go unparsed again: shed the optional blocks
        for k in ("other_configs", "kernel_time_ms_per_step", "package_default", "path_roofline"):
            out.pop(k, None)
            line = json.dumps(out, separators=(",", ":"))
            if len(line) < LINE_LIMIT:
                break
    return line


def emit(result):
    """stdout carries exactly ONE line, the compact one (the driver's record lost the 25 KB line of round 4, and a long earlier
    stdout line could push the compact one out of whatever window the driver keeps).  Everything else goes to
    bench_detail.json beside this script, to gpurun_out/bench_detail.json (merged back from a GPU box) and to stderr."""
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(result, f, indent=1)
        except OSError as e:   # a read-only checkout must not cost the bench line
            sys.stderr.write("bench.py: %s not written (%s)\n" % (path, e))
    sys.stderr.write("bench_detail: " + json.dumps(result) + "\n")
    sys.stderr.flush()
    print(compact_line(result), flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec through torch.distributed.run with N ranks on 127.0.0.1 (the
    command line the driver uses) and return its exit status.  Refuses -- non-zero, nothing printed on stdout -- when fewer
    than N devices are visible: a `--gpus 8` request is never answered by a smaller run."""
    import subprocess
    n = args.gpus
    if n < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if not args.dry_spawn:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if args.rehearse_on_one_device and have >= 1:
            have = n
        if have < n:
            sys.stderr.write("bench.py: --gpus %d requested but %d ROCm device(s) visible on this node; refusing to run a "
                             "smaller configuration in its place\n" % (n, have))
            return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (native libraries that write to stdout behind Python's back)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def dry_rank(args, rank, world):
    """--dry-spawn, one rank: the rendezvous, the process group (gloo: no device) and the metadata exchange of
    decode_sharded(local_shard=True) -- every rank contributes (F, wireframes held, padded width) -- then rank 0 prints
    the planned shards.  Exercises everything of an N-rank launch except the GPU work."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    with stdout_to_stderr():      # ("[Gloo] Rank r is connected to ..." goes to stdout of every rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfgE = args.config == "E"
    W = args.wireframes_per_gpu or (32 if cfgE else (min(128, max(1, 1024 // world)) if world > 1 else 1))
    n_local = config_e_edge_counts(world * W)[rank * W:(rank + 1) * W] if cfgE else [args.edges] * W
    meta = torch.tensor([max(n_local), len(n_local), 1024 if cfgE else args.edges], dtype=torch.int64)
    allmeta = torch.empty(3 * world, dtype=torch.int64)
    dist.all_gather_into_tensor(allmeta, meta)
    allmeta = allmeta.view(world, 3).tolist()
    ranks_seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(ranks_seen)
    if rank == 0:
        print(json.dumps({"dry_spawn": True, "n_gpus": world, "backend": dist.get_backend(), "ranks_in_group": int(ranks_seen.item()),
                          "shard_sizes": [int(m[1]) for m in allmeta], "global_F": max(int(m[0]) for m in allmeta),
                          "global_batch": sum(int(m[1]) for m in allmeta), "config": args.config,
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}))
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="B", choices=["B", "E"],
                    help="B: 256-edge wireframes (BASELINE configs 2/3, the headline); E: ours-perspective.yml with "
                         "num_lines=1024, ragged 64..1024-edge wireframes (BASELINE config 5)")
    ap.add_argument("--wireframes-per-gpu", type=int, default=0,
                    help="0 = 1 (config B at N = 1) / 128 (config B at N > 1: config 3's share) / 32 (config E)")
    ap.add_argument("--edges", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=16, help="wireframes per micro-batch (0 = all)")
    ap.add_argument("--chunk-max-seqs", type=int, default=8192, help="sequences per micro-batch of several wireframes")
    ap.add_argument("--chunk-seqs", type=int, default=0, help="sequences per intra-wireframe group (0 = off)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent HIP streams for the micro-batches")
    ap.add_argument("--attn-algo", type=int, default=0, help="ff_attention kernel: 0 auto, 1 LDS-shared, 2 wave")
    ap.add_argument("--gemm-tuning", default="", help="min_units,two_per_cu_units,fix_tenths[,small_max_rows] of ff_set_gemm_tuning")
    ap.add_argument("--x3-min-rows", type=int, default=0,
                    help="3 x bf16 projections (fp32-accurate, bf16 matrix cores) on launches with at least this many rows. "
                         "The HEADLINE is measured with 0 (every product on the f32 matrix cores, dtype f32); the package "
                         "default is measured as well and reported under 'package_default'")
    ap.add_argument("--split-kind", default="", choices=["", "bf16x3", "fp16x2"],
                    help="how the split projections of the package-default line split an fp32 operand (default: the package's)")
    ap.add_argument("--no-fuse-ln", action="store_true", help="standalone LayerNorm launches (A/B of FF_FUSE_LAYERNORM)")
    ap.add_argument("--no-dedup", action="store_true", help="decode every padding-anchor row like the reference does")
    ap.add_argument("--sync-every", type=int, default=1, help="host stop-rule check period in steps (0 = never; package default 1)")
    ap.add_argument("--cpu-anchors", type=int, default=0,
                    help="anchor sequences in the CPU baseline (0 = the first 128: ~22 s of CPU work; 256 = the FULL wireframe, ~44 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="torch threads of the CPU oracle (0 = physical cores, at most 32)")
    ap.add_argument("--cpu-timeout", type=int, default=200, help="wall-clock cap of the CPU baseline [s]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-x3-line", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed PMC passes instead of two rocprofv3 --pmc child passes of this run")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C128 / E32 / D / A lines of the N = 1 run")
    ap.add_argument("--other-steps", type=int, default=1, help="timed passes of each `other_configs` entry")
    ap.add_argument("--other-list", default="C128,E32,D,A64",
                    help="which `other_configs` entries to measure (all: C128,E32,D,A,D64,A64; the default keeps the run short)")
    ap.add_argument("--other-profiled", default="C128",
                    help="which `other_configs` entries also get a profiled pass (their own kernel roofline); 'all' = every entry")
    ap.add_argument("--ln-fuse-max-rows", type=int, default=0, help="LayerNorm folded into the projections up to this many rows (0: 12288)")
    ap.add_argument("--plain-multi", action="store_true",
                    help="N > 1: time the plain per-rank model(batch) + all-gather (the round-2 form) instead of decode_sharded")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the N > 1 code path (process group, decode_sharded, collectives) even with WORLD_SIZE = 1: a "
                         "self-check of that path on a one-GPU box, not a benchmark configuration")
    ap.add_argument("--dry-spawn", action="store_true",
                    help="launch plumbing only, no GPU: spawn --gpus ranks the way a real run does, build a gloo group, agree the "
                         "shard sizes with the collectives decode_sharded(local_shard=True) uses and print them as one JSON line")
    ap.add_argument("--no-json-gather", action="store_true", help="N > 1: skip the timed face-loop JSON gather (second figure)")
    ap.add_argument("--rehearse-on-one-device", action="store_true",
                    help="N > 1 on a ONE-GPU box: all ranks share device 0 and gloo is the collective backend (RCCL refuses two ranks "
                         "on one device).  Runs every line of the N-rank code path; the figures are NOT a measurement (the line says so)")
    args = ap.parse_args()
    t_start = time.perf_counter()

    # ---- launch contract: `--gpus N` is what runs, or nothing does ---------------------------------------------------
    # Launched by torch.distributed.run (WORLD_SIZE set): WORLD_SIZE must equal --gpus.  Launched bare with --gpus N > 1:
    # this process spawns the N ranks itself through torch.distributed.run (after checking that N devices are visible) and
    # exits with their status -- it never falls through to a one-GPU run that would print `n_gpus: 1` for `--gpus 8`.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_spawn):
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or run `python bench.py "
                         "--gpus %d` bare and let it spawn its ranks)" % (args.gpus, world, args.gpus, args.gpus))
    if args.dry_spawn:
        return dry_rank(args, rank, world)
    multi = world > 1 or args.force_dist   # the distributed code path
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the decode path has no CPU fallback)")
    rehearsal = bool(args.rehearse_on_one_device and world > 1)
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node by contract: the host-side control group of decode_sharded
        # gloo's C++ side reports its connections on STDOUT of every rank ("[Gloo] Rank r is connected to ..."): stdout is the
        # driver's one JSON line, so the groups (RCCL's and the host-side control twin of decode_sharded) are made with fd 1
        # pointed at stderr
        with stdout_to_stderr():
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            from faceformer_amd.dist import ensure_control_group
            ensure_control_group(dist)

    from faceformer_amd.config import load_cfg
    from faceformer_amd.dist import decode_sharded, gather_predictions
    from faceformer_amd.hip import lib as L
    from faceformer_amd.hip import ops as _ops
    from faceformer_amd.models import SurfaceFormer, SurfaceFormer_Parallel
    from faceformer_amd.models.common import X3_MIN_ROWS_DEFAULT
    from faceformer_amd.synth import make_extra_mask, make_state_dict, make_wireframes, state_dict_spec

    lib = L.load()
    _ops.set_attention_algo(args.attn_algo)
    if args.gemm_tuning:
        _ops.set_gemm_tuning(*[int(v) for v in args.gemm_tuning.split(",")])

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def to_dev(b):
        return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}

    def parallel_model(cfg_file, L_lines, recipe="default", wseed=0):
        cfg = load_cfg(os.path.join(ROOT, "configs", cfg_file), ["model.num_lines", str(L_lines)])
        T = cfg.model.max_face_length
        model = SurfaceFormer_Parallel(**cfg.model)
        spec = state_dict_spec("parallel", L_lines, T, cfg.model.num_model, cfg.model.num_feedforward,
                               cfg.model.num_encoder_layers, cfg.model.num_decoder_layers)
        model.load_state_dict(make_state_dict(spec, recipe, wseed))
        model = model.eval().to(dev)
        model.chunk_wireframes, model.chunk_max_seqs = args.chunk, args.chunk_max_seqs
        model.chunk_seqs, model.num_streams = args.chunk_seqs, args.streams
        model.sync_every = args.sync_every
        model.x3_min_rows = args.x3_min_rows
        if args.split_kind:
            model.split_kind = args.split_kind
        if args.no_dedup:
            model.decode_flags = model.decode_flags & ~L.FF_DEDUP_PAD_ANCHORS
        if args.no_fuse_ln:
            model.decode_flags = model.decode_flags & ~L.FF_FUSE_LAYERNORM
        apply_knobs(model)
        return model, cfg, T

    def apply_knobs(mod):
        mod.ln_fuse_max_rows = args.ln_fuse_max_rows
        if os.environ.get("FF_BENCH_LN_EPILOGUE"):     # (A/B: "1" = the split products normalise in their epilogue, "0" = rows first)
            mod.x3_ln_in_epilogue = os.environ["FF_BENCH_LN_EPILOGUE"] == "1"

    def steps_executed(pred):   # pred [N, F, T] or [N, T]
        p = pred.reshape(-1, pred.size(-1))
        nz = (p[:, 1:] != 0).any(dim=0)
        return int(nz.nonzero().max().item()) + 1 if bool(nz.any()) else 0

    def effective_clock(step_fn, sec_per_pass):
        """GHz the SIMDs ran at while `step_fn` passes were executing: a one-wave probe kernel on a second stream reads the
        shader-clock counter against the constant 100 MHz counter (ff_clock_probe_*) during ~80 % of a ~0.6 s run of passes."""
        side = torch.cuda.Stream()
        long_pass = sec_per_pass > 0.5            # (128 wireframes per GPU: 5.6 s per pass -- one pass, the probe inside it)
        n = 1 if long_pass else max(2, int(0.6 / max(sec_per_pass, 1e-4)) + 1)
        spin_us = min(0.8 * n * sec_per_pass, 2.0) * 1e6   # ff_clock_probe_launch takes at most 5 s
        try:
            if not long_pass:
                step_fn()
            torch.cuda.synchronize()
            L.check(lib.ff_clock_probe_launch(ctypes.c_double(spin_us), side.cuda_stream), "ff_clock_probe_launch")
            for _ in range(n):
                step_fn()
            torch.cuda.synchronize()
            g, u = ctypes.c_double(0.0), ctypes.c_double(0.0)
            L.check(lib.ff_clock_probe_read(ctypes.byref(g), ctypes.byref(u), side.cuda_stream), "ff_clock_probe_read")
            return float(g.value)
        except L.HipExtensionError as e:          # a measurement hook must never cost the bench line
            sys.stderr.write("bench.py: effective clock not measured (%s)\n" % e)
            torch.cuda.synchronize()
            return 0.0

    bracket_us = 0.0
    if rank == 0 and not args.no_roofline:
        b = ctypes.c_double(0.0)
        L.check(lib.ff_profile_bracket_us(512, ctypes.byref(b), torch.cuda.current_stream().cuda_stream), "ff_profile_bracket_us")
        bracket_us = float(b.value)

    # ================================================================================================================
    # main line
    # ================================================================================================================
    cfgE = args.config == "E"
    sharded_c = multi and not cfgE and not args.plain_multi
    W = args.wireframes_per_gpu or (32 if cfgE else (min(128, max(1, 1024 // world)) if sharded_c else 1))
    L_lines = 1024 if cfgE else args.edges
    model, cfg, T = parallel_model("ours-perspective.yml" if cfgE else "ours.yml", L_lines)
    seeds = [rank * W + i for i in range(W)]
    if cfgE:
        all_n = config_e_edge_counts(world * W)
        n_local = all_n[rank * W:(rank + 1) * W]
    else:
        all_n = [args.edges] * (world * W)
        n_local = [args.edges] * W
    batch = to_dev(make_wireframes(n_local, L_lines, T, "parallel", seeds=seeds))
    F_local = max(n_local)

    def step_plain():
        with torch.no_grad():
            out = model(dict(batch))
        pred = out["predict"]
        if multi:
            if cfgE and pred.size(1) < max(all_n):   # ragged shards: pad the anchor dimension to the global F
                pad = torch.zeros((pred.size(0), max(all_n) - pred.size(1), pred.size(2)), dtype=pred.dtype, device=dev)
                pred = torch.cat([pred, pad], dim=1)
            pred = gather_predictions(pred, dist)
        return pred

    def step_sharded():   # every rank holds ONLY its own wireframes; result = the whole batch on every rank
        with torch.no_grad():
            return decode_sharded(model, dict(batch), dist, local_shard=True)["predict"]

    step = step_sharded if (multi and not args.plain_multi) else step_plain
    dt, pred = timed(step, fence, args.warmup, args.steps)
    if multi:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # decode steps actually executed (the reference semantics run T-1 with these weights)
    local = pred[rank * W:(rank + 1) * W] if multi else pred
    steps_done = steps_executed(local)
    # decoded edges = pointer selections of the REAL anchor sequences (n_w per wireframe); the reference
    # additionally decodes F - n_w identical padding-anchor rows per wireframe, reported separately
    sel_per_step = sum(all_n) * steps_done
    value = sel_per_step * args.steps / dt
    stats = getattr(model, "last_decode_stats", None) or {}

    if cfgE:
        hist = {str(k): all_n.count(k) for k in E_SIZES}
        workload = ("BASELINE config 5: configs/ours-perspective.yml model.num_lines=1024: %d synthetic wireframes per GPU with edge "
                    "counts drawn from %s p=%s (numpy default_rng(%d); this run: %s), F=max n anchor rows per wireframe x %d greedy "
                    "steps, default-xavier synthetic weights" % (W, list(E_SIZES), list(E_PROBS), E_SEED, hist, steps_done))
    elif sharded_c:
        workload = ("BASELINE config 3: configs/ours.yml model.num_lines=%d, synthetic %d-edge wireframes sharded by wireframe: "
                    "%d per GPU x %d GPUs = %d in the batch (config 3 is 1024 over 8 GPUs = 128 per GPU; this series keeps 128 per "
                    "GPU), F=%d anchor sequences x %d greedy steps each, default-xavier synthetic weights; every rank passes only "
                    "its own wireframes to dist.decode_sharded(local_shard=True): batch-global F, all-reduced stop counters, global "
                    "stop rule, RCCL all-gather of the int32 tokens inside the timed step"
                    % (args.edges, args.edges, W, world, W * world, args.edges, steps_done))
    else:
        workload = ("BASELINE config 2: configs/ours.yml model.num_lines=%d: %d synthetic %d-edge wireframe(s) per GPU, "
                    "F=%d anchor sequences x %d greedy steps, default-xavier synthetic weights"
                    % (args.edges, W, args.edges, args.edges, steps_done))
    result = {
        "metric": "decoded edges/sec (greedy face-loop decode, pointer selections/s), 256-edge wireframes"
                  if not cfgE else "decoded edges/sec (greedy face-loop decode, pointer selections/s), ragged 64-1024-edge wireframes",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "wireframes_per_gpu": W, "edges": args.edges if not cfgE else "64..1024",
                   "max_face_length": T, "decode_steps": steps_done,
                   "parallelism": "wireframe-sharded x%d, RCCL all-gather of predictions" % world},
        "wireframes_per_s": world * W * args.steps / dt,
        "ms_per_wireframe_per_gpu": 1e3 * dt / args.steps / W,
    }
    if multi:
        try:
            rv = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001 - the version string is informational
            rv = None
        result["rccl_ranks"] = dist.get_world_size()          # what the collective library saw, not what --gpus asked for
        result["collective_backend"] = "%s (RCCL on ROCm)" % dist.get_backend() if not rehearsal else "gloo"
        if rehearsal:
            result["rehearsal"] = ("NOT A MEASUREMENT: --rehearse-on-one-device, %d ranks share device 0 and gloo moves the device "
                                   "tensors; run to exercise the N-rank code path on a one-GPU box" % world)
        result["rccl_version"] = rv
    if sharded_c:
        result["scaling_series"] = {
            "per_gpu_workload": "%d wireframes of %d edges (config 3's per-GPU share)" % (W, args.edges),
            "n1_reference": "other_configs.C128 of the N = 1 line (same per-GPU workload on one GPU, no collective)",
            "note": "the N = 1 line's own `value` is BASELINE config 2 (ONE wireframe per call), a different workload: a batch of "
                    "128 wireframes fills the per-kernel latency of the single-wireframe decode (DESIGN.md 5)"}
    if cfgE:
        rows = W * F_local
        result["sequence_rows"] = {
            "reference_rows_per_gpu": rows, "decoded_sequences_per_gpu": stats.get("decoded_seqs"),
            "real_anchor_sequences_per_gpu": sum(n_local),
            "padding_rows_not_decoded_per_gpu": rows - (stats.get("decoded_seqs") or rows),
            "note": "rows f >= n_w of a wireframe are identical padding-anchor sequences (reference model_para.py:204-205); "
                    "one is decoded per wireframe and copied"}
    falg = sum(alg_flops_per_wireframe(n, T) for n in n_local)
    result["path_roofline"] = path_roofline(falg, dt / args.steps)

    if multi and sharded_c:
        # the one-wireframe-per-GPU weak line (what N = 1 measures as its headline), plain model(batch) + all-gather
        one = to_dev(make_wireframes([args.edges], L_lines, T, "parallel", seeds=[rank]))

        def step_one():
            with torch.no_grad():
                return gather_predictions(model(dict(one))["predict"], dist)
        dt1, p1 = timed(step_one, fence, 1, 3)
        tt = torch.tensor([dt1], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt1 = float(tt.item())
        result["weak_one_wireframe_per_gpu"] = {
            "value": world * args.edges * steps_executed(p1[rank:rank + 1]) * 3 / dt1, "unit": "edges/s", "ms_per_step": 1e3 * dt1 / 3,
            "steps": 3, "warmup": 1, "workload": "ONE %d-edge wireframe per GPU (BASELINE config 2 on every GPU), local stop rule, "
                                                 "all-gather of the predictions" % args.edges}

    if multi and sharded_c and not args.no_json_gather:
        # second figure of the north-star: the predicted face loops as JSON on every rank -- decode + face parsing of the
        # rank's own wireframes (host, reference trainer.py:118-136,181-208) + the length-prefixed RCCL all-gather of the bytes
        from faceformer_amd.dist import decode_to_face_json
        nrec = [0, 0]

        def step_json():
            with torch.no_grad():
                recs = decode_to_face_json(model, dict(batch), dist, local_shard=True)
            nrec[0], nrec[1] = len(recs), sum(len(r) for r in recs)
            return recs
        dtj, _ = timed(step_json, fence, 1, 2)
        tt = torch.tensor([dtj], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dtj = float(tt.item())
        result["face_json_gather"] = {
            "value": sel_per_step * 2 / dtj, "unit": "edges/s", "ms_per_step": 1e3 * dtj / 2, "steps": 2, "warmup": 1,
            "records": nrec[0], "json_bytes": nrec[1],
            "workload": "the same sharded decode, then face parsing of the rank's own wireframes on the host and the all-gather of "
                        "the per-wireframe JSON records (u32 length prefix + utf-8, one padded uint8 all_gather_into_tensor over RCCL)"}

    if not multi and args.x3_min_rows == 0 and not args.no_x3_line:
        # second line: the package default (large decoder projections as fp32-accurate 3 x bf16 products)
        model.x3_min_rows = X3_MIN_ROWS_DEFAULT
        dt3, _ = timed(step, fence, max(1, args.warmup), args.steps)
        roof3 = None
        if rank == 0 and not args.no_roofline:
            def once3():
                with torch.no_grad():
                    model(dict(batch))
            roof3 = x3_roofline(profile_once(lib, L, once3), 1e3 * dt3 / args.steps,
                                with_traffic=(not cfgE and args.edges == 256 and W == 1),
                                kind=getattr(model, "split_kind", "bf16x3"))
            ghz3 = effective_clock(once3, dt3 / args.steps)
            roof3["effective_clock_ghz"] = ghz3
            roof3["frac_at_effective_clock"] = roof3["frac"] * NOMINAL_GHZ / ghz3 if ghz3 > 0 else None
        model.x3_min_rows = 0
        step()      # (re-binds the engine without the bf16 planes for the profiling leg below)
        fence()
        result["package_default"] = {
            "value": sel_per_step * args.steps / dt3, "unit": "edges/s", "ms_per_step": 1e3 * dt3 / args.steps,
            "x3_min_rows": X3_MIN_ROWS_DEFAULT, "split_kind": getattr(model, "split_kind", "bf16x3"),
            "note": "package default: decoder projections of launches with >= %d rows (q|k|v; linear1 from 7/4 x, the 512-column "
                    "ones from 11/4 x that) as split products on the 16-bit matrix cores (split_kind fp16x2: two fp16 terms, three "
                    "products; bf16x3: three bf16 terms, six products) with the LayerNorms folded in, fp32-accurate; NOT the headline"
                    % X3_MIN_ROWS_DEFAULT,
            "path_roofline": path_roofline(falg, dt3 / args.steps)}
        if roof3 is not None:
            result["package_default"]["roofline"] = roof3

    if rank == 0 and not args.no_roofline:
        def once():
            with torch.no_grad():
                model(dict(batch))
        prof = profile_once(lib, L, once)
        # HBM bytes per launch of the dominant kernel come from the committed PMC passes
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 fetch correction):
        # bench.py cannot run the profiler around itself.
        traffic, traffic_src = None, None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
        if not cfgE and args.edges == 256 and W == 1:
            why = "--no-live-traffic"
            if not args.no_live_traffic and not multi:
                traffic, traffic_src = live_traffic()
                why = traffic_src if traffic is None else None
            if traffic is None and cands:      # the committed PMC passes of the same command (labelled as such)
                with open(cands[-1]) as f:
                    tj = json.load(f)
                traffic = tj["hbm_bytes_per_launch"]
                traffic_src = "%s (committed passes; live measurement: %s)" % (os.path.relpath(cands[-1], ROOT), why)
        result["roofline"], extra = gemm_roofline(prof, 1e3 * dt / args.steps, bracket_us, traffic, traffic_src)
        result.update(extra)
        ghz = effective_clock(once, dt / args.steps)
        result["roofline"]["effective_clock_ghz"] = ghz
        result["roofline"]["frac_at_effective_clock"] = result["roofline"]["frac"] * NOMINAL_GHZ / ghz if ghz > 0 else None

    cpu_thread = None
    if rank == 0 and not multi and not args.no_cpu_baseline:
        # The oracle runs in child processes with hard wall-clock caps.  Threads: SURVEY 8(d) says all physical cores, but
        # torch's CPU eager path gets SLOWER beyond ~32 threads for these operator sizes (measured on the MI355X host:
        # 256 threads 1.4 edges/s, 64 threads 133, 32 threads 165-245; 128 threads: no result within 3 x the 32-thread time
        # in every run of round 5).  The baseline sample runs at min(physical cores, 32) threads (`cores`); a 16-anchor
        # sample is timed at that setting and at all physical cores afterwards (`threads_sweep`).
        import subprocess
        n_cpu = n_local[0] if not cfgE else min(n_local)      # config E: the smallest wireframe of the shard
        seed_cpu = seeds[0] if not cfgE else seeds[n_local.index(n_cpu)]
        phys = physical_cores()
        threads = args.cpu_threads if args.cpu_threads > 0 else min(phys, 32)
        threads = max(1, min(threads, os.cpu_count() or 1))
        pinfo = " ".join(torch.__config__.parallel_info().split())[:400]

        def child_code(k, threads):
            return (
                "import sys, time, json, torch\n"
                "sys.path.insert(0, %r)\n"
                "torch.set_num_threads(%d)\n"
                "from oracle import refpath\n"
                "from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec\n"
                "spec = state_dict_spec('parallel', %d, %d, %d, %d, %d, %d)\n"
                "sd = make_state_dict(spec, 'default', 0)\n"
                "one = make_wireframes(%d, %d, %d, 'parallel', seeds=[%d])\n"
                "trace = {}\n"
                "t0 = time.perf_counter()\n"
                "ref = refpath.parallel_forward_eval(sd, one, num_head=%d, anchor_limit=%s, trace=trace)\n"
                "tc = time.perf_counter() - t0\n"
                "lg = torch.stack(trace['logits'])\n"                       # steps x B x S masked logits of the oracle
                "v = torch.sort(lg, dim=2, descending=True).values\n"
                "live = lg > torch.finfo(torch.float32).min\n"
                "scale = (lg.abs() * live).amax(dim=(1, 2))\n"
                "print(json.dumps({'t': tc, 'predict': ref['predict'][0].tolist(), 'margin': (v[:, :, 0] - v[:, :, 1]).tolist(), "
                "'scale': scale.tolist()}))\n"
                % (ROOT, threads, L_lines, T, cfg.model.num_model, cfg.model.num_feedforward,
                   cfg.model.num_encoder_layers, cfg.model.num_decoder_layers, n_cpu, L_lines, T, seed_cpu,
                   cfg.model.num_head, "None" if k >= n_cpu else str(k)))

        code_a = (
            "import sys, time, json, torch\n"
            "sys.path.insert(0, %r)\n"
            "torch.set_num_threads(%d)\n"
            "from oracle import refpath\n"
            "from faceformer_amd.synth import make_state_dict, make_wireframes, state_dict_spec\n"
            "sd = make_state_dict(state_dict_spec('seq2seq', 110, 259), 'gain4', 0)\n"
            "one = make_wireframes(64, 110, 259, 'seq2seq', seeds=[3])\n"
            "t0 = time.perf_counter()\n"
            "ref = refpath.seq2seq_forward_eval(sd, one, num_head=8)\n"
            "tc = time.perf_counter() - t0\n"
            "p = ref['predict'][0]\n"
            "print(json.dumps({'t': tc, 'steps': int((p[1:] != 0).nonzero().max()) + 1}))\n" % (ROOT, min(threads, 8)))
        cpu_raw = {"sweep": {}, "tried": [], "rec": None, "k": None, "threads": threads, "rec_a": None}

        def cpu_job():
            """The baseline sample (and config A's sizes) on a host thread BESIDE the GPU measurement of `other_configs.C128`
            only: C128 is GPU-bound (5.6 s per pass, unchanged beside 32 busy host threads), the launch-bound entries are not
            (profiles/r05: D 111.7 -> 133.2 ms beside 32 threads, E32 1318 -> 1788 ms beside the 128-thread sweep child), so
            they run first and the thread sweep runs after everything else, alone."""
            th_full = cpu_raw["threads"]
            # (bounded sample: the first 128 anchor sequences are ~22 s at 32 threads and end with the GPU's C128 passes; the full
            # wireframe -- 44 s, --cpu-anchors 256 -- was the default until round 5 and kept the whole run 20 s longer)
            for k in ([args.cpu_anchors] if args.cpu_anchors > 0 else [min(n_cpu, 128), 32]):
                k = max(1, min(k, n_cpu))
                try:
                    cpu_raw["rec"] = run_cpu_child(child_code(k, th_full), args.cpu_timeout, th_full)
                    cpu_raw["k"] = k
                    break
                except (subprocess.TimeoutExpired, ValueError, IndexError) as e:
                    cpu_raw["tried"].append("%d anchors: no result within %ds (%s)" % (k, args.cpu_timeout, type(e).__name__))
            if not cfgE:
                try:   # one sequence of <= 258 rows: more than 8 threads only add synchronisation (64 threads: 14/s)
                    cpu_raw["rec_a"] = run_cpu_child(code_a, 120, min(th_full, 8))
                except (subprocess.TimeoutExpired, ValueError, IndexError):
                    pass

        def cpu_sweep():
            """SURVEY 8(d) says all physical cores; torch's CPU eager path is slower beyond ~32 threads for these operator
            sizes.  The first 16 anchor sequences at 32 threads and at all physical cores, alone on the host (after every GPU
            measurement); the second setting gets 2 x the time of the first -- slower than that it cannot win."""
            if not (args.cpu_threads <= 0 and phys > cpu_raw["threads"] and not args.cpu_anchors):
                return
            ks = max(1, min(16, n_cpu))
            t_first = None
            for th in (cpu_raw["threads"], phys):
                cap = 60 if t_first is None else max(8, int(2 * t_first) + 3)
                t0_ = time.perf_counter()
                try:
                    r_ = run_cpu_child(child_code(ks, th), cap, th)
                    pr_ = torch.tensor(r_["predict"], dtype=torch.int64)
                    st_ = int((pr_[:, 1:] != 0).any(dim=0).nonzero().max().item()) + 1
                    cpu_raw["sweep"][str(th)] = ks * st_ / r_["t"]
                except (subprocess.TimeoutExpired, ValueError, IndexError):
                    cpu_raw["sweep"][str(th)] = "no result within %d s" % cap
                if t_first is None:
                    t_first = time.perf_counter() - t0_

        import threading
        cpu_thread = threading.Thread(target=cpu_job, name="cpu-baseline", daemon=True)

    def start_cpu():
        # beside the GPU-bound entry (C128) only -- see cpu_job
        if cpu_thread is not None and not cpu_thread.is_alive() and cpu_thread.ident is None:
            cpu_thread.start()

    # ================================================================================================================
    # N = 1: the other BASELINE configurations in the same run (a few timed passes each)
    # ================================================================================================================
    if not multi and not cfgE and W == 1 and not args.no_other_configs:
        other = {}
        K2 = max(1, args.other_steps)
        want = set(args.other_list.split(","))
        prof_all = args.other_profiled == "all"
        want_prof = set(args.other_profiled.split(","))

        def profiled(name):
            return not args.no_roofline and (prof_all or name in want_prof)

        def par_entry(name, m2, b2, n2, T2, what):
            def st():
                with torch.no_grad():
                    return m2(dict(b2))["predict"]
            d2, p2 = timed(st, fence, 1, K2)
            sd2 = steps_executed(p2)
            ent = {"workload": what, "value": sum(n2) * sd2 * K2 / d2, "unit": "edges/s", "ms_per_step": 1e3 * d2 / K2,
                   "ms_per_wireframe": 1e3 * d2 / K2 / len(n2), "wireframes_per_s": len(n2) * K2 / d2, "steps": K2, "warmup": 1,
                   "decode_steps": sd2, "dtype": "f32",
                   "path_roofline": path_roofline(sum(alg_flops_per_wireframe(n, T2) for n in n2), d2 / K2)}
            if profiled(name):
                roof, ex = gemm_roofline(profile_once(lib, L, st), 1e3 * d2 / K2, bracket_us)
                ent["roofline"] = roof
                ent.update(ex)
            if not args.no_x3_line:
                m2.x3_min_rows = X3_MIN_ROWS_DEFAULT
                d3, _ = timed(st, fence, 1, K2)
                r3 = x3_roofline(profile_once(lib, L, st), 1e3 * d3 / K2, kind=getattr(m2, "split_kind", "bf16x3")) if profiled(name) else None
                m2.x3_min_rows = 0
                ent["package_default"] = {"value": sum(n2) * sd2 * K2 / d3, "unit": "edges/s", "ms_per_step": 1e3 * d3 / K2,
                                             "ms_per_wireframe": 1e3 * d3 / K2 / len(n2), "roofline": r3,
                                             "path_roofline": path_roofline(sum(alg_flops_per_wireframe(n, T2) for n in n2), d3 / K2),
                                             "note": "package default (x3_min_rows = %d), fp32-accurate; not the headline form" % X3_MIN_ROWS_DEFAULT}
            other[name] = ent

        # D / A: the single-sequence model (SurfaceFormer), one wireframe, 258 steps with the gain-4 parity weights
        wbytes = decoder_weight_bytes()
        for name, cfg_file, n1, wseed, wfseed, mask_seed, what in (
                ("D", "seq2seq+coedge.yml", 216, 1, 5, 11,
                 "BASELINE config 4: configs/seq2seq+coedge.yml (num_lines 216, label_seq_length 259), one 216-edge synthetic "
                 "wireframe, extra pointer mask (co-edge style, ~20 %% of the edges) OR-ed into the padding mask inside the pointer "
                 "kernel, gain-4 synthetic weights (the weights / wireframe / mask of golden seq_full_D216_extramask)"),
                ("A", "seq2seq.yml", 64, 0, 3, None,
                 "BASELINE config 1's sizes on the GPU: configs/seq2seq.yml (num_lines 110, label_seq_length 259), one 64-edge "
                 "synthetic wireframe, gain-4 synthetic weights (golden seq_full_A64_gain4)")):
            if name not in want and (name + "64") not in want:
                continue
            c1 = load_cfg(os.path.join(ROOT, "configs", cfg_file))
            L1, T1 = c1.model.num_lines, c1.model.label_seq_length
            m1 = SurfaceFormer(**c1.model)
            m1.load_state_dict(make_state_dict(state_dict_spec("seq2seq", L1, T1), "gain4", wseed))
            m1 = m1.eval().to(dev)
            m1.x3_min_rows = 0
            apply_knobs(m1)
            K1 = max(3, K2)
            if name in want:
                b1 = make_wireframes([n1], L1, T1, "seq2seq", seeds=[wfseed])
                if mask_seed is not None:
                    b1["extra_mask"] = make_extra_mask(dict(kind="seq2seq", extra_mask_seed=mask_seed), b1)
                b1 = to_dev(b1)

                def st1():
                    with torch.no_grad():
                        return m1(dict(b1))["predict"]
                d1, p1 = timed(st1, fence, 1, K1)
                sd1 = steps_executed(p1)
                sec = d1 / K1
                fa = alg_flops_per_wireframe(n1, T1, F=1)
                ent = {"workload": what, "value": sd1 / sec, "unit": "edges/s", "ms_per_step": 1e3 * sec, "ms_per_wireframe": 1e3 * sec,
                       "steps": K1, "warmup": 1, "decode_steps": sd1, "dtype": "f32", "path_roofline": path_roofline(fa, sec),
                       "bounds": {
                           "mfma": {"alg_tflop": fa / 1e12, "bound_ms": 1e3 * fa / (PEAK_F32_MFMA_TFLOPS * 1e12),
                                    "frac": fa / (PEAK_F32_MFMA_TFLOPS * 1e12) / sec},
                           "weight_stream": {"bytes_per_wireframe": sd1 * wbytes, "bound_ms": 1e3 * sd1 * wbytes / (PEAK_HBM_TBS * 1e12),
                                             "frac": sd1 * wbytes / (PEAK_HBM_TBS * 1e12) / sec,
                                             "note": "decoder + project weights (%.1f MB fp32) streamed once per decode step at the %.0f TB/s "
                                                     "HBM rate (SURVEY 8d; they also fit the 256 MB Infinity Cache)" % (wbytes / 1e6, PEAK_HBM_TBS)}}}
                if profiled(name):
                    roof, ex = gemm_roofline(profile_once(lib, L, st1), 1e3 * sec, bracket_us)
                    ent["roofline"] = roof
                    ent.update(ex)
                    ent["launches_per_decode_step"] = sum(ex["kernel_launches_per_step"].values()) / max(1, sd1)
                other[name] = ent
            # The throughput form of the same configuration: 64 wireframes per call (the reference's forward_eval takes a batch;
            # its stop rule waits for EVERY wireframe's EOS).  A one-wireframe decode is 53 dependent launches per step at the
            # ~7 us floor of a dependent one-tile launch (accepted); 64 sequences per step fill the same launches.
            if (name + "64") in want:
                b64 = make_wireframes([n1] * 64, L1, T1, "seq2seq", seeds=list(range(wfseed, wfseed + 64)))
                if mask_seed is not None:
                    b64["extra_mask"] = make_extra_mask(dict(kind="seq2seq", extra_mask_seed=mask_seed), b64)
                b64 = to_dev(b64)

                def st64():
                    with torch.no_grad():
                        return m1(dict(b64))["predict"]
                K64 = K2
                d64, p64 = timed(st64, fence, 1, K64)
                sd64 = steps_executed(p64)
                sec64 = d64 / K64
                fa64 = 64 * alg_flops_per_wireframe(n1, T1, F=1)
                e64 = {"workload": what + " -- 64 such wireframes (seeds %d..%d) in ONE call" % (wfseed, wfseed + 63),
                       "value": 64 * sd64 / sec64, "unit": "edges/s", "ms_per_step": 1e3 * sec64, "ms_per_wireframe": 1e3 * sec64 / 64,
                       "steps": K64, "warmup": 1, "decode_steps": sd64, "dtype": "f32", "path_roofline": path_roofline(fa64, sec64),
                       "bounds": {"mfma": {"alg_tflop": fa64 / 1e12, "bound_ms": 1e3 * fa64 / (PEAK_F32_MFMA_TFLOPS * 1e12),
                                           "frac": fa64 / (PEAK_F32_MFMA_TFLOPS * 1e12) / sec64},
                                  "weight_stream": {"bytes": sd64 * wbytes, "bound_ms": 1e3 * sd64 * wbytes / (PEAK_HBM_TBS * 1e12),
                                                    "frac": sd64 * wbytes / (PEAK_HBM_TBS * 1e12) / sec64}}}
                if profiled(name + "64"):
                    roof, ex = gemm_roofline(profile_once(lib, L, st64), 1e3 * sec64, bracket_us)
                    e64["roofline"] = roof
                    e64.update(ex)
                other[name + "64"] = e64
                del b64
            del m1
        # E32: config 5's per-GPU share
        if "E32" in want:
            mE, cE, TE = parallel_model("ours-perspective.yml", 1024)
            nE = config_e_edge_counts(8 * 32)[:32]
            bE = to_dev(make_wireframes(nE, 1024, TE, "parallel", seeds=list(range(32))))
            par_entry("E32", mE, bE, nE, TE,
                      "BASELINE config 5's per-GPU share: 32 ragged wireframes, edge counts %s (configs/ours-perspective.yml, "
                      "model.num_lines=1024, max_face_length %d), padding anchors de-duplicated, width-bucketed micro-batches"
                      % ({str(k): nE.count(k) for k in E_SIZES}, TE))
            other["E32"]["decoded_sequences"] = (getattr(mE, "last_decode_stats", None) or {}).get("decoded_seqs")
            del mE, bE

        start_cpu()
        # C128: config 3's per-GPU share on this GPU (the model of the main line, a batch of 128 wireframes)
        if "C128" in want:
            bC = to_dev(make_wireframes([args.edges] * 128, L_lines, T, "parallel", seeds=list(range(128))))
            par_entry("C128", model, bC, [args.edges] * 128, T,
                      "BASELINE config 3's per-GPU share: 128 synthetic %d-edge wireframes in one call (configs/ours.yml, "
                      "model.num_lines=%d), micro-batches of %d wireframes, default-xavier weights" % (args.edges, L_lines, args.chunk))
            del bC
        result["other_configs"] = other

    if cpu_thread is not None:
        start_cpu()
        cpu_thread.join()
        cpu_sweep()
        wf_local = local[seeds.index(seed_cpu)]
        tried, sweep, rec, k, threads = cpu_raw["tried"], cpu_raw["sweep"], cpu_raw["rec"], cpu_raw["k"], cpu_raw["threads"]
        if rec is not None:
            ref_pred = torch.tensor(rec["predict"], dtype=torch.int64)
            tc = rec["t"]
            ref_steps = int((ref_pred[:, 1:] != 0).any(dim=0).nonzero().max().item()) + 1
            # default-init weights give the reference EXACT logit ties in a few sequences (tests/golden:
            # 174 of 9216 selections); a tie may legitimately resolve differently under another fp32
            # summation order, after which that sequence's later tokens differ too
            eq = (ref_pred[:k].to(dev) == wf_local[:k]).cpu()
            same = bool(eq.all())
            seq_same = float(eq.all(dim=1).float().mean())
            # where the GPU's tokens first leave the oracle's: (sequence, step), the ORACLE's own top-2 margin there and
            # the parity tolerance of that step (tests/test_parity_golden.py: 1e-3 * max(1, max|logit| / 40))
            div = []
            for s_ in (~eq.all(dim=1)).nonzero().flatten().tolist():
                j = int((~eq[s_]).nonzero().min().item())            # token index; decode step = j - 1
                step_ = j - 1
                tol = 1e-3 * max(1.0, rec["scale"][step_] / 40.0)
                div.append({"seq": s_, "step": step_, "ref_margin": rec["margin"][step_][s_], "tol": tol,
                            "ref_token": int(ref_pred[s_, j]), "gpu_token": int(wf_local[s_, j])})
            div.sort(key=lambda d: (d["step"], d["seq"]))
            what = ("all %d anchor sequences" % k) if k >= n_cpu else ("first %d of %d anchor sequences" % (k, n_cpu))
            result["cpu_baseline"] = {
                "value": k * ref_steps / tc, "unit": "edges/s", "cores": threads, "kind": "port",
                "sample": "%s of one %d-edge wireframe of the workload, all %d steps, oracle/refpath.py (torch %s CPU eager "
                          "fp32; %d threads on %d physical cores / %d hardware threads): %.1f s%s"
                          % (what, n_cpu, ref_steps, torch.__version__, threads, phys, os.cpu_count() or 1, tc,
                             ("; earlier attempts: " + "; ".join(tried)) if tried else ""),
                "threads_sweep": {"sample": "first 16 anchor sequences, edges/s by thread count, alone on the host", **sweep} if sweep else None,
                "parallel_info": pinfo, "tokens_identical_to_gpu": same, "sequences_identical_to_gpu": seq_same,
                "divergent_sequences": len(div),
                "first_divergence": div[0] if div else None,
                "every_first_divergence_within_2tol_of_a_tie": all(d["ref_margin"] <= 2 * d["tol"] for d in div),
                "first_divergences": div[:8],
                "note": "a sequence may leave the oracle's tokens only where the oracle's own top-2 margin is below the parity "
                        "tolerance (an exact or near tie resolved under another fp32 summation order); the parity tests require "
                        "equality wherever the margin exceeds 2 x tol",
                "ran_beside": "the GPU measurements of other_configs (one host thread of the bench process; the headline and the "
                              "package-default line were finished before the first CPU child started)",
            }
            result["speedup_vs_cpu"] = value / result["cpu_baseline"]["value"]

        else:
            result["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": threads, "kind": "port",
                                      "sample": "; ".join(tried)}
        if cpu_raw["rec_a"] is not None:
            rec = cpu_raw["rec_a"]
            result["cpu_baseline_config_a"] = {
                "value": rec["steps"] / rec["t"], "unit": "edges/s", "cores": min(threads, 8), "kind": "port",
                "sample": "configs/seq2seq.yml sizes (L=110, T=259), one 64-edge wireframe, gain-4 weights (the workload of "
                          "other_configs.A), all %d executed steps: %.1f s" % (rec["steps"], rec["t"])}

    if rank == 0:
        result["bench_seconds"] = time.perf_counter() - t_start
        emit(result)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
