import sqlite3, sys
db, tiles, Ms = sys.argv[1], [int(x) for x in sys.argv[2].split(",")], [int(x) for x in sys.argv[3].split(",")]
c = sqlite3.connect(db)
g = [r for r in c.execute("select name, start, end from kernels order by start").fetchall() if 'gemm' in r[0]]
i = 0
for M in Ms:
    for (K, N) in ((512, 512), (512, 1536), (1024, 512)):
        line = "M=%5d K=%4d N=%4d |" % (M, K, N)
        for t in tiles:
            ds = sorted((r[2] - r[1]) / 1e3 for r in g[i * 6:(i + 1) * 6])[1:-1]
            us = sum(ds) / len(ds)
            line += " t%-2d %6.1fus %5.1fTF |" % (t, us, 2 * M * N * K / us / 1e6)
            i += 1
        print(line)
