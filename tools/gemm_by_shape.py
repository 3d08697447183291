"""profiles/r06/gemm_by_shape_in_decode.txt: the six projection shapes of a full decoder layer, per decode step, IN the decode
(from a rocprofv3 kernel trace of bench.py's config B: tools/run_trace.sh) with the isolated-loop figure of the same launch beside
it (tools/bench_gemm_ln.py's table, optional).

    python tools/gemm_by_shape.py <rocpd db> <t,t,...> [bench_gemm_ln table]

A decode step's launches come in a fixed order (ff_engine.hip: decoder_pass); layers 1..nd-2 run over all t*256 rows:
    q|k|v (N 1536, LN consumer)  self-attn  out-proj (N 512, stats)  cross-q (N 512, LN consumer)  cross-attn
    cross-out (N 512, stats)  linear1 (N 1024, LN consumer, ReLU)  linear2 (K 1024, stats)
The table takes layers 1..4 (layer 0 has the cached q|k|v, layer 5 is pruned to the newest position) and prints their mean.
"""
import re
import sqlite3
import sys

SHAPES = [("q|k|v", 1536, 512, "ln-in"), ("out-proj", 512, 512, "stats"), ("cross-q", 512, 512, "ln-in"),
          ("cross-out", 512, 512, "stats"), ("linear1", 1024, 512, "ln-in"), ("linear2", 512, 1024, "stats")]


def steps_of(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    steps, cur, decodes = [], [], []
    for n, s, e in rows:
        cur.append((n, s, e))
        if 'pointer_reduce' in n or 'pointer_kernel' in n:
            steps.append(cur); cur = []
        if 'finalize' in n and 'chunk' in n:
            decodes.append(steps); steps = []; cur = []
    return decodes[-1]


def layer_gemms(step):
    """-> list over layers 1..4 of the six (kernel, us) of that layer.  A full layer is the run gemm, attn, gemm, gemm, attn, gemm,
    gemm, gemm; layer 0 starts with the newest rows' q|k|v (a small launch) and has the same pattern, so layers are cut at
    every 'gemm attn gemm gemm attn gemm gemm gemm' match in order and the first (layer 0) and last (pruned) are dropped."""
    kinds = "".join("a" if "attention" in n else ("g" if "gemm" in n else "o") for n, _s, _e in step)
    out = []
    for m in re.finditer("gaggaggg", kinds):
        i = m.start()
        idx = [i, i + 2, i + 3, i + 5, i + 6, i + 7]
        out.append([(re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", step[j][0])[:34], (step[j][2] - step[j][1]) / 1e3) for j in idx])
    return out[1:5]


def isolated(path):
    tab = {}
    if not path:
        return tab
    for line in open(path):
        f = line.replace("|", " ").split()
        if len(f) == 7 and f[0].isdigit():
            M, K, N = int(f[0]), int(f[1]), int(f[2])
            tab[(M, K, N)] = dict(plain=float(f[3]), stats=float(f[4]), **{"ln-in": float(f[5])})
    return tab


def main():
    d = steps_of(sys.argv[1])
    ts = [int(x) for x in sys.argv[2].split(",")]
    iso = isolated(sys.argv[3] if len(sys.argv) > 3 else None)
    print("config B (256 sequences): rows = 256 t; per shape: mean us of layers 1..4 in the decode, TF/s, 64x128 tiles per CU "
          "(or 64x64 tiles per CU for the 64x64 family), isolated-loop us / TF/s of the same launch form")
    print("%3s %6s %-10s %5s %5s %-34s %8s %7s %6s | %8s %7s" % ("t", "rows", "shape", "N", "K", "kernel", "us", "TF/s", "t/CU", "iso us", "TF/s"))
    for t in ts:
        if t > len(d):
            continue
        layers = layer_gemms(d[t - 1])
        if len(layers) < 4:
            print("%3d: layer pattern not found (%d matches)" % (t, len(layers)))
            continue
        M = 256 * t
        for k, (name, N, K, form) in enumerate(SHAPES):
            us = sum(L[k][1] for L in layers) / len(layers)
            kern = layers[0][k][0]
            fl = 2.0 * M * N * K
            dma = "dma" in kern
            tpc = ((M + 63) // 64) * ((N + (127 if dma else 63)) // (128 if dma else 64)) / 256.0
            i = iso.get((M, K, N), {}).get(form)
            print("%3d %6d %-10s %5d %5d %-34s %8.1f %7.1f %6.2f | %8s %7s" % (
                t, M, name, N, K, kern, us, fl / us / 1e6, tpc, "%.1f" % i if i else "-", "%.1f" % (fl / i / 1e6) if i else "-"))
        tot = sum(sum(x[1] for x in L) for L in layers) / len(layers)
        flt = sum(2.0 * M * N * K for _n, N, K, _f in SHAPES)
        print("%3d %6d %-10s %45s %8.1f %7.1f" % (t, M, "layer", "", tot, flt / tot / 1e6))


if __name__ == "__main__":
    main()
