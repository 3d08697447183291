"""Per-kernel totals of one or more PMC counters from a rocprofv3 rocpd sqlite file.
    python tools/pmc_per_kernel.py results.db [out.md]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = {}
for k, cn, v in rows:
    k = re.sub(r"\(.*", "", re.sub(r"^void\s+", "", k.replace("(anonymous namespace)::", "")))[:70]
    a = agg.setdefault((k, cn), [0, 0.0])
    a[0] += 1
    a[1] += float(v)
lines = ["| kernel | counter | dispatches | sum | mean per dispatch |", "|---|---|---|---|---|"]
for (k, cn), (n, s) in sorted(agg.items(), key=lambda kv: (-kv[1][1] if True else 0)):
    lines.append("| %s | %s | %d | %.6g | %.6g |" % (k, cn, n, s, s / n))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
