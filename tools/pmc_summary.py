"""Summarise rocprofv3 --pmc output (rocpd sqlite): mean counter value per kernel.
    python tools/pmc_summary.py results.db [kernel-substring]"""
import re
import sqlite3
import sys

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
print("columns:", cols)
rows = c.execute("select * from counters_collection").fetchall()
idx = {n: i for i, n in enumerate(cols)}
kname = [n for n in cols if "kernel" in n and "name" in n] or [n for n in cols if n == "name"]
cname = [n for n in cols if "counter" in n and "name" in n]
val = [n for n in cols if n in ("value", "counter_value")]
agg = {}
for r in rows:
    k = r[idx[kname[0]]]
    if sub and sub not in k:
        continue
    k = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", ""))
    key = (k, r[idx[cname[0]]])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(r[idx[val[0]]])
for (k, cn), (n, s) in sorted(agg.items()):
    print("%-60s %-32s n=%4d mean=%.4g" % (k[:60], cn, n, s / n))
