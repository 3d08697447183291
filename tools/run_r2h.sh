set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k attention > gpurun_out/r2h/pytest.log 2>&1; tail -2 gpurun_out/r2h/pytest.log
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
echo "B: $(run) $(run) $(run)"
timeout 600 python tools/attn_sweep.py 2>&1 | tail -8
