"""Per decode step: wall span and per-category kernel time from a rocprofv3 kernel trace of bench.py."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
steps, cur, decodes = [], [], []
for n, s, e in rows:
    cur.append((n, s, e))
    if 'pointer_reduce' in n or 'pointer_kernel' in n:
        steps.append(cur); cur = []
    if 'finalize' in n and 'chunk' in n:
        decodes.append(steps); steps = []; cur = []
d = decodes[-1]
tot = 0
print("t  wall_us  gemm attn ln other n")
for i, st in enumerate(d):
    t0, t1 = st[0][1], st[-1][2]
    g = sum(e - s for n, s, e in st if 'gemm' in n)
    a = sum(e - s for n, s, e in st if 'attention' in n)
    l = sum(e - s for n, s, e in st if 'layernorm' in n)
    o = sum(e - s for n, s, e in st) - g - a - l
    tot += t1 - t0
    print("%2d %8.1f %7.1f %6.1f %6.1f %5.1f %d" % (i + 1, (t1 - t0) / 1e3, g / 1e3, a / 1e3, l / 1e3, o / 1e3, len(st)))
print("total ms", tot / 1e6)
if len(sys.argv) > 2:  # kernel-by-kernel listing of the given steps (1-based)
    import re
    for t in [int(x) for x in sys.argv[2].split(",")]:
        print("---- step", t)
        prev = None
        for n, s, e in d[t - 1]:
            short = re.sub(r"\(anonymous namespace\)::|void |\(.*$", "", n)[:60]
            print("%-60s %7.1f us  gap %5.1f" % (short, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
            prev = e
