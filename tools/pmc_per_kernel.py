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
# duration of the dispatches UNDER THIS PASS (the profiled run is slower than the traced one: effective clock = GRBM_GUI_ACTIVE /
# 8 XCDs / this duration, never the kernel-trace duration of another run), when the view carries the timestamps
try:
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    if "start" in cols and "end" in cols:
        idc = "dispatch_id" if "dispatch_id" in cols else None
        q = "select kernel_name, start, end%s from counters_collection" % ((", " + idc) if idc else "")
        seen = set()
        for row in c.execute(q):
            k = re.sub(r"\(.*", "", re.sub(r"^void\s+", "", row[0].replace("(anonymous namespace)::", "")))[:70]
            key = (k, row[3]) if idc else (k, row[1], row[2])
            if key in seen:
                continue
            seen.add(key)
            a = agg.setdefault((k, "DURATION_NS"), [0, 0.0])
            a[0] += 1
            a[1] += float(row[2] - row[1])
except sqlite3.Error:
    pass
lines = ["| kernel | counter | dispatches | sum | mean per dispatch |", "|---|---|---|---|---|"]
for (k, cn), (n, s) in sorted(agg.items(), key=lambda kv: (-kv[1][1] if True else 0)):
    lines.append("| %s | %s | %d | %.6g | %.6g |" % (k, cn, n, s, s / n))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
