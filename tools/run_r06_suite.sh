# Run ON THE GPU BOX: the whole GPU suite with the margins file, the step-1 probe, the default bench line.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
rm -f $O/parity_margins.txt
FF_PARITY_MARGINS=$PWD/$O/parity_margins.txt timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_full.log 2>&1; tail -5 $O/pytest_gpu_full.log
timeout 300 python tools/step1_probe.py > $O/step1_untraced.txt 2>&1; cat $O/step1_untraced.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_stderr.txt; cut -c1-1500 $O/bench_line.json
cp bench_detail.json $O/bench_detail.json 2>/dev/null
