# Run ON THE GPU BOX: same-box A/B of library builds on the PACKAGE DEFAULT (split products on): config B and 128 wireframes per call.
# usage: bash tools/run_r06_default_ab.sh <other.so> ...
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
X3=${FF_AB_X3_ROWS:-1024}
runb() { timeout 300 python bench.py --x3-min-rows $X3 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
runc() { timeout 600 python bench.py --x3-min-rows $X3 --wireframes-per-gpu 128 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.0f' % d['value'])"; }
for i in 1 2; do
  echo "in-tree: B $(runb) ms  C128 $(runc) edges/s"
  for o in "$@"; do echo "$o: B $(FF_HIP_LIB=$PWD/$o runb) ms  C128 $(FF_HIP_LIB=$PWD/$o runc) edges/s"; done
done
