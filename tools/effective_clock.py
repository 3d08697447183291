"""Effective shader clock UNDER LOAD of the path's dominant kernels, with the board's own power / clock read-outs beside it:
    python tools/effective_clock.py > profiles/r05/effective_clock.md          (run on the GPU box)

For each kernel (3 x bf16 split projection at two large shapes, the f32 LDS-DMA projection, the
K/V-resident cross-attention kernel) and each operand fill (random / zeros: the chip clocks to its POWER budget, and zero
operands toggle no matrix-core data lines), the launches run back to back for ~2 s on torch's stream while
  * a one-wave probe kernel on a SECOND stream (ff_clock_probe_launch: s_memtime cycles / s_memrealtime wall time) measures
    the clock the SIMDs actually ran at during the middle of that loop, and
  * a child process samples `rocm-smi --showpower --showclocks --json` every ~100 ms.
Printed per row: TF/s (fp32-equivalent flops / GPU time of the median ~20-ms chunk of launches; the share of chunks within 5 % of
that median says how much of the loop the host kept the queue full), effective GHz of the probe, the rate re-priced at that clock
(fraction of peak x 2.4 / GHz), board power [W] and the sclk the SMI tool reports (min / mean / max of the samples).
"""
import ctypes
import json
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faceformer_amd.hip import lib as L  # noqa: E402
from faceformer_amd.hip import ops  # noqa: E402

NOMINAL_GHZ = 2.4
PEAK_F32 = 157.3
PEAK_X3 = 2500.0 / 6.0


class Smi:
    """Samples board power and sclk in a CHILD PROCESS (a shell loop around rocm-smi writing JSON lines to a file): a sampler
    thread of this interpreter would share the GIL with the launch loop and slow the launches it is meant to observe."""

    def __init__(self):
        self.power, self.sclk = [], []
        self.tool = None
        for cand in ("/opt/rocm/bin/rocm-smi", "rocm-smi"):
            try:
                subprocess.run([cand, "--version"], capture_output=True, timeout=20)
                self.tool = cand
                break
            except (OSError, subprocess.TimeoutExpired):
                continue
        self.path = "/tmp/ff_smi_%d.jsonl" % os.getpid()
        self.proc = None

    def start(self):
        if not self.tool:
            return
        open(self.path, "w").close()
        self.proc = subprocess.Popen(["bash", "-c", "while true; do %s --showpower --showclocks --json >> %s 2>/dev/null; echo >> %s; "
                                      "sleep 0.05; done" % (self.tool, self.path, self.path)])

    def stop(self):
        if self.proc is None:
            return
        self.proc.terminate()
        try:
            self.proc.wait(timeout=10)
        except subprocess.TimeoutExpired:
            self.proc.kill()
        for line in open(self.path):
            line = line.strip()
            if not line.startswith("{"):
                continue
            try:
                d = json.loads(line)
            except ValueError:
                continue
            card = d.get("card0") or (list(d.values())[0] if d else {})
            for k, v in card.items():
                kl = k.lower()
                try:
                    if "power" in kl and "(w)" in kl:
                        self.power.append(float(v))
                    elif kl.startswith("sclk clock speed"):
                        self.sclk.append(float(str(v).strip("()").lower().replace("mhz", "")))
                except ValueError:
                    pass
        os.remove(self.path)


def mmm(v):
    return "%.0f / %.0f / %.0f" % (min(v), sum(v) / len(v), max(v)) if v else "n/a"


def run_case(name, fn, flops, peak, seconds=2.0):
    lib = L.load()
    side = torch.cuda.Stream()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        fn()
    torch.cuda.synchronize()
    one = max((time.perf_counter() - t0) / 16, 1e-6)
    iters = max(8, int(seconds / one))
    smi = Smi()
    smi.start()
    # GPU-side timing in chunks of ~20 ms (an event per chunk boundary): the MEDIAN chunk rate is what the kernel does when the
    # launch loop keeps the queue full; chunks far below it are host stalls (the interpreter was descheduled), counted in `steady`
    ck = max(4, int(0.02 / one))
    nchunks = max(4, iters // ck)
    iters = ck * nchunks
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nchunks + 1)]
    evs[0].record()
    launched = False
    for c in range(nchunks):
        for _ in range(ck):
            fn()
        evs[c + 1].record()
        if not launched and c >= nchunks // 4:      # the probe covers the middle half of the loop
            L.check(lib.ff_clock_probe_launch(ctypes.c_double(0.5 * seconds * 1e6), side.cuda_stream), "ff_clock_probe_launch")
            launched = True
    torch.cuda.synchronize()
    smi.stop()
    ghz, us = ctypes.c_double(0.0), ctypes.c_double(0.0)
    L.check(lib.ff_clock_probe_read(ctypes.byref(ghz), ctypes.byref(us), side.cuda_stream), "ff_clock_probe_read")
    secs = sorted(evs[c].elapsed_time(evs[c + 1]) * 1e-3 for c in range(nchunks))
    med = secs[len(secs) // 2]
    steady = sum(1 for v in secs if v <= 1.05 * med) / float(len(secs))
    tf = flops * ck / med / 1e12
    frac = tf / peak
    print("| %s | %d | %.1f | %.2f | %.3f | %.3f | %.3f | %s | %s |" % (
        name, iters, tf, steady, frac, ghz.value, frac * NOMINAL_GHZ / ghz.value if ghz.value > 0 else float("nan"), mmm(smi.power),
        mmm(smi.sclk)))
    sys.stdout.flush()


def main():
    dev = "cuda"
    lib = L.load()
    # idle clock first (nothing else queued)
    side = torch.cuda.Stream()
    L.check(lib.ff_clock_probe_launch(ctypes.c_double(2e5), side.cuda_stream), "ff_clock_probe_launch")
    g, u = ctypes.c_double(0.0), ctypes.c_double(0.0)
    L.check(lib.ff_clock_probe_read(ctypes.byref(g), ctypes.byref(u), side.cuda_stream), "ff_clock_probe_read")
    print("# Effective shader clock under load (tools/effective_clock.py; probe = s_memtime cycles / s_memrealtime wall time of one")
    print("# wave on a second stream during the middle half of a ~2 s launch loop; nominal %.1f GHz)" % NOMINAL_GHZ)
    print()
    print("idle probe (no other work queued): %.3f GHz over %.0f us" % (g.value, u.value))
    print()
    print("| kernel, shape, operands | launches | TF/s (fp32-eq, median 20-ms chunk) | chunks within 5 % of the median | frac of peak @2.4 GHz | effective GHz | frac of peak @effective clock | board power W (min / mean / max) | SMI sclk MHz (min / mean / max) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for fill in ("random", "zeros"):
        for M, K, N in ((16384, 512, 1536), (9216, 1024, 512)):   # (launches of >= 60 us: the Python loop stays ahead of the GPU;
            # the config-B-sized launches are probed in situ by bench.py: package_default.roofline.effective_clock_ghz)
            a = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev)
            if fill == "zeros":
                a.zero_(); w.zero_(); b.zero_()
            out = torch.empty(M, N, device=dev)
            planes = ops.split_weight(w)
            st = torch.cuda.current_stream().cuda_stream
            pa, pw, pb, po, pp = a.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), planes.data_ptr()
            # (prebuilt arguments, straight ctypes calls: a few microseconds of host time per launch, so that the 26-us launches
            # of the config-B-sized shape stay GPU-bound)
            run_case("gemm_x3_kernel (3 x bf16) %dx%d->%d %s" % (M, K, N, fill),
                     lambda: lib.ff_gemm_x3(pa, K, None, 0, pp, pb, None, 0, po, N, M, N, K, 0, st), 2.0 * M * N * K, PEAK_X3)
            planes2 = ops.split_weight(w, "fp16x2")       # round 6: two fp16 terms, three products (the package default)
            pp2 = planes2.data_ptr()
            run_case("gemm_x3_kernel (2 x fp16) %dx%d->%d %s" % (M, K, N, fill),
                     lambda: lib.ff_gemm_x2h(pa, K, None, 0, pp2, pb, None, 0, po, N, M, N, K, 0, st), 2.0 * M * N * K, 2.0 * PEAK_X3)
            if True:
                run_case("gemm_dma_f32_kernel %dx%d->%d %s" % (M, K, N, fill),
                         lambda: lib.ff_gemm_f32(pa, K, None, 0, pw, K, pb, None, 0, po, N, M, N, K, 0, 11, st), 2.0 * M * N * K, PEAK_F32)
            del a, w, b, out, planes
        # K/V-resident cross-attention of config B at t = 36: 256 sequences x 36 positions on S = 260 keys, 8 heads of 64
        F_, t, S, H, E = 256, 36, 260, 8, 512
        q = torch.randn(t * F_, E, device=dev)
        kv = torch.randn(S, 2 * E, device=dev)
        if fill == "zeros":
            q.zero_(); kv.zero_()
        kvl = torch.tensor([S], device=dev, dtype=torch.int32)
        run_case("attention_resident_kernel F=256 t=36 S=260 %s" % fill,
                 lambda: ops.attention(q, kv[:, :E], kv[:, E:], num_groups=1, num_heads=H, nq=F_ * t, nk=S, q_group_stride=F_,
                                       q_inner=F_, q_outer_stride=F_, k_group_stride=S, k_stride=1, kv_len=kvl, scale=0.125),
                 4.0 * F_ * t * S * E, PEAK_F32)


if __name__ == "__main__":
    main()
