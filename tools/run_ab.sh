# Run ON THE GPU BOX (via gpurun): same-box A/B of builds of the library (FF_HIP_LIB overrides the in-tree one).
# usage: bash tools/run_ab.sh <other.so> [<other2.so> ...]; prints seq2seq (config A, 1 wireframe) and config-B times.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "${FF_AB_TESTS:-gemm or linear}" > gpurun_out/ab/pytest.log 2>&1; tail -2 gpurun_out/ab/pytest.log
runb() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runs() { FF_SEQ_ONLY_A=1 timeout 300 python tools/time_seq2seq.py 2>&1 | grep seq2seq | awk '{print $5}'; }
for i in 1 2; do
  echo "in-tree: B $(runb) ms  seq2seq $(runs) ms"
  for o in "$@"; do echo "$o: B $(FF_HIP_LIB=$o runb) ms  seq2seq $(FF_HIP_LIB=$o runs) ms"; done
done
