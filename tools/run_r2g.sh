set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2g
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
tail -4 gpurun_out/r2g/pytest.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
echo "default(sync 4 lagged) $(run) $(run)   sync0 $(run --sync-every 0) $(run --sync-every 0)  sync1 $(run --sync-every 1)"
timeout 600 python tools/time_seq2seq.py 2>&1 | tail -2
