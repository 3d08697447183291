"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel launches, total / average duration.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [out.md]
"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        short = name.replace("(anonymous namespace)::", "")
        short = re.sub(r"^void\s+", "", short)
        short = re.sub(r"\(.*", "", short)
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (k[:90], a[0], a[1] / 1e3, a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    lines.append("| TOTAL | %d | %.3f | | | | 100 |" % (sum(a[0] for a in agg.values()), tot / 1e3))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
