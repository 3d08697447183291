cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in x3p f32; do
for s in "1024 512 512" "2048 512 512" "4096 512 512" "6144 512 512" "9216 512 512" "2048 1536 512" "4096 1536 512" "4096 512 1024" "4096 1024 512"; do
  rm -rf /tmp/pt; rocprofv3 --kernel-trace -d /tmp/pt -o t -- python tools/x3_probe.py $s 10 $mode > /dev/null 2>&1
  python - "$mode" $s <<PY
import sqlite3, sys
c = sqlite3.connect("/tmp/pt/t_results.db")
rows = c.execute("select name, end-start from kernels").fetchall()
d = [t for n, t in rows if ("gemm_x3p" in n or "gemm_streamk" in n or "gemm_persist" in n)]
d = d[1:]
M, N, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
us = sum(d) / len(d) / 1e3
print("%s M=%d N=%d K=%d: %.1f us  %.1f TF/s" % (sys.argv[1], M, N, K, us, 2.0 * M * N * K / us / 1e6))
PY
done; done
