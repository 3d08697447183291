cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
runb() { timeout 300 python bench.py --x3-min-rows 1024 --no-cpu-baseline --no-roofline --no-x3-line --no-other-configs --steps 8 --warmup 2 "$@" 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f' % d['ms_per_step'])"; }
for i in 1 2 3; do echo "auto: $(runb) ms   algo4 (x2h wherever planes exist): $(runb --attn-algo 4) ms   algo3 (resident f32 wherever possible): $(runb --attn-algo 3) ms"; done
