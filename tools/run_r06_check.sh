# Run ON THE GPU BOX: whole GPU suite, then the default bench line (re-entry check of the restored tree).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
rm -f $O/parity_margins.txt
FF_PARITY_MARGINS=$PWD/$O/parity_margins.txt timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
T0=$(date +%s); timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench_stderr.txt; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
cut -c1-1500 $O/bench_line.json
