import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
tot = sum(e - s for _, s, e in rows)
# union length
cur_s, cur_e, union = None, None, 0
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("kernels %d  sum of durations %.1f ms  union (wall busy) %.1f ms  overlap factor %.2f  span %.1f ms" % (
    len(rows), tot / 1e6, union / 1e6, tot / union, (rows[-1][2] - rows[0][1]) / 1e6))
