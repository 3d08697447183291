# Run ON THE GPU BOX (via gpurun): gemm_panel_kernel on / off (FF_NO_PANEL=1) in the same build: op tests, config B, seq2seq.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/panel
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "gemm or linear" > gpurun_out/panel/pytest.log 2>&1; tail -2 gpurun_out/panel/pytest.log
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runs() { timeout 300 python tools/time_seq2seq.py 2>&1 | grep seq2seq | awk "{print \$1, \$3, \$5}" | tr "\n" " "; }
for i in 1 2; do
  echo "panel: B $(runb) ms  seq2seq $(runs) ms"
  echo "no panel: B $(FF_NO_PANEL=1 runb) ms  seq2seq $(FF_NO_PANEL=1 runs) ms"
done 2>&1 | tee gpurun_out/panel/ab.txt
