set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attention" > gpurun_out/ab/pytest.log 2>&1; tail -2 gpurun_out/ab/pytest.log
python tools/attn_sweep.py --seq 2>&1 | tail -22
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
for i in 1 2; do echo "B $(runb) ms"; done
timeout 600 python tools/time_seq2seq.py 2>&1 | grep "call=1 "
