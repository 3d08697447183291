"""Per decode step wall time from a rocprofv3 kernel trace of bench.py when a step runs as G concurrent launch chains
(chunk_seqs groups on several streams): step k is over when its G-th pointer_reduce launch has ended.
usage: step_walls.py <results.db> <G>"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
G = int(sys.argv[2])
rows = c.execute("select name, start, end from kernels order by end").fetchall()
ends, decodes = [], []
for n, s, e in rows:
    if 'pointer_reduce' in n:
        ends.append(e)
    if 'finalize' in n and 'chunk' in n:
        if ends:
            decodes.append(ends)
        ends = []
d = decodes[-1]
done = [d[i] for i in range(G - 1, len(d), G)]
first = min(s for n, s, e in rows if s > done[0] - 5e6 and 'init_tokens' in n and s < done[0])
prev, tot = first, 0.0
print("t  wall_us")
for i, e in enumerate(done):
    print("%2d %8.1f" % (i + 1, (e - prev) / 1e3))
    tot += e - prev
    prev = e
print("total ms %.3f" % (tot / 1e6))
